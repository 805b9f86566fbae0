/*
 * cli_shim.cpp -- TEST INFRASTRUCTURE ONLY.  The GPU-side entry points of the C ABI (include/bowtie_amd.h) answered by the
 * host build of the device automatons (bt_emu.cpp), as a library for LD_PRELOAD under the bowtie-amd binary: lets the
 * binary's own host logic -- option handling, batching, the second pass for reads with many hits, pairs, read dumps,
 * tallies, output order -- run against the reference's outputs where no GPU exists.  Nothing in the product loads it; without
 * it (and without a GPU) bowtie-amd stops at "could not load index".  The searches it performs are the emulator's.
 */
#include "bt_emu.cpp"
#include <deque>
#include <mutex>

static std::mutex g_emu_mutex;        /* the emulator loads the 2-bit reference on first use: one search at a time */

struct bt_index { void* emu = nullptr; bool mirror = false; int variant = 1; };
struct bt_ctx { const bt_index* ix = nullptr; bt_policy pol; bool best = false; std::deque<void*> done; };

extern "C" int bt_has_pe_v1(void) { return 1; }

extern "C" int bt_index_load(const char* base, int need_mirror, int offrate_override, int device, bt_index** out)
{
	(void)device;
	if (!base || !out) return BT_ERR_ARG;
	*out = nullptr;
	const int variant = bt_host_index_variant(base);
	if (variant < 0) return BT_ERR_IO;
	void* e = nullptr;
	try { e = emu_index_load(base, need_mirror, offrate_override); } catch (const std::exception&) { return BT_ERR_FORMAT; }
	if (!e) {
		/* the loader's own verdict (an index of 2^32-1 rows or more under the 32-bit build: BT_ERR_ROWS64, on which the binary
		 * starts bowtie-amd-l) */
		BtIndexHost probe;
		int rc = BT_ERR_FORMAT;
		try { rc = bt_host_index_load(base, true, offrate_override, &probe); } catch (const std::exception&) { rc = BT_ERR_FORMAT; }
		return rc != BT_OK ? rc : BT_ERR_FORMAT;
	}
	bt_index* ix = new bt_index();
	ix->emu = e; ix->mirror = need_mirror != 0; ix->variant = variant;
	*out = ix;
	return BT_OK;
}
extern "C" void bt_index_info_get(const bt_index* idx, bt_index_info* info)
{
	memset(info, 0, sizeof(*info));
	const BtIndexHost& h = ((EmuIndex*)idx->emu)->h[0];
	info->len = h.len > 0xffffffffull ? 0xffffffffu : (uint32_t)h.len; info->n_pat = h.nPat; info->n_frag = h.nFrag; info->ftab_chars = (uint32_t)h.ftabChars;
	info->off_rate = (uint32_t)h.offRate; info->z_off = (uint32_t)h.zOff;
	info->has_mirror = idx->mirror ? 1 : 0;
	info->variant = idx->variant | (h.swapped ? BT_INDEX_SWAPPED : 0);
}
extern "C" const char* bt_index_refname(const bt_index* idx, uint32_t t)
{
	const BtIndexHost& h = ((EmuIndex*)idx->emu)->h[0];
	return t < h.refnames.size() ? h.refnames[t].c_str() : nullptr;
}
extern "C" uint32_t bt_index_reflen(const bt_index* idx, uint32_t t)
{
	const BtIndexHost& h = ((EmuIndex*)idx->emu)->h[0];
	return t < h.plen.size() ? (uint32_t)h.plen[t] : 0;
}
extern "C" void bt_index_free(bt_index* idx) { if (idx) { emu_index_free(idx->emu); delete idx; } }
extern "C" int bt_index_load_reference(bt_index* ix) { return ix ? BT_OK : BT_ERR_ARG; }       /* emu_align_pairs loads it on first use */

extern "C" int bt_ctx_create(const bt_index* idx, const bt_policy* pol, void* stream, bt_ctx** out)
{
	(void)stream;
	if (!idx || !pol || !out) return BT_ERR_ARG;
	*out = nullptr;
	bt_ctx* c = new bt_ctx();
	c->ix = idx; c->pol = *pol; c->best = pol->best != 0 || pol->pe_v1 != 0;
	c->pol.best = c->best ? 1 : 0;
	if (c->best) { BfProgram P; const int rc = bt_host_compile_best(*pol, &P); if (rc != BT_OK) { delete c; return rc; } }
	else { BtProgram P; const int rc = bt_host_compile_program(*pol, &P); if (rc != BT_OK) { delete c; return rc; } }
	*out = c;
	return BT_OK;
}
extern "C" void bt_ctx_destroy(bt_ctx* c) { delete c; }

/* the return code of bt_align_batch / bt_align_pairs (bt_api.cpp): the worst thing that happened to a read */
static int worst_status(const bt_hit_batch* out, uint32_t n)
{
	int worst = BT_OK;
	for (uint32_t i = 0; i < n; i++) {
		if (out->status[i] & BT_ST_TOOSHORT) worst = BT_ERR_READ_SHORT;
		else if ((out->status[i] & (BT_ST_OVERFLOW | BT_ST_MMPOOL)) && worst == BT_OK) worst = BT_ERR_OVERFLOW;
	}
	return worst;
}

/* SHIM_NULL_SEARCH=1 (scripts/cli_host_bench.py): no search at all -- three reads in four get one made-up alignment
 * with up to two mismatches -- so that a run of the binary times its own host pipeline (reader, batching, formatter,
 * writer) and nothing else.  The alignments mean nothing. */
static bool null_search() { static const bool v = getenv("SHIM_NULL_SEARCH") != nullptr; return v; }
static int null_align(const bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out)
{
	const BtIndexHost& h = ((EmuIndex*)c->ix->emu)->h[0];
	const uint32_t nref = (uint32_t)h.plen.size();
	uint32_t pool = 0;
	for (uint32_t i = 0; i < in->n_reads; i++) {
		const uint32_t L = in->len[i], x = (in->seed[i] | 1u) * 2654435761u, t = x % nref;      /* the read's own seed: the same answer in any batch */
		out->status[i] = 0; out->n_hits[i] = 0;
		if ((x >> 28 & 3u) == 3u || L == 0 || h.plen[t] <= L) continue;
		bt_hit& H = out->hits[(size_t)i * out->hit_cap];
		memset(&H, 0, sizeof(H));
		H.tidx = t; H.toff = (x >> 7) % ((uint32_t)h.plen[t] - L); H.fw = (uint8_t)(x >> 5 & 1u); H.nmm = (uint16_t)(x >> 3 & 3u) % 3u;
		H.cost = (uint16_t)(H.nmm * 30u); H.stratum = (uint8_t)H.nmm;
		if (H.nmm && out->mm_pool && pool + H.nmm <= out->mm_pool_cap) {
			H.mm_off = pool;
			for (uint32_t k = 0; k < H.nmm; k++) out->mm_pool[pool++] = (uint16_t)(((k + 1u) * L / 4u) | ((x >> (9 + 2 * k) & 3u) << 12));
		} else H.nmm = 0;
		out->n_hits[i] = 1;
	}
	out->mm_pool_used = pool;
	return BT_OK;
}

extern "C" int bt_align_batch(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, bt_op_counts* counts)
{
	if (!c || !in || !out) return BT_ERR_ARG;
	if (in->n_reads == 0) return BT_OK;
	if (null_search()) return null_align(c, in, out);
	uint32_t maxLen = 1;
	for (uint32_t i = 0; i < in->n_reads; i++) if (in->len[i] > maxLen) maxLen = in->len[i];
	/* arenas no read of the batch can outgrow (the library reaches the same through its second pass) */
	const uint32_t L = maxLen < 64 ? 64 : maxLen;
	const uint32_t frCap = L + 8u, entCap = (L * (L + 3u) / 2u + 64u + 7u) & ~7u;
	std::lock_guard<std::mutex> lock(g_emu_mutex);
	const int rc = emu_align_batch(c->ix->emu, &c->pol, in, out, counts, 64, frCap, c->best ? 0u : entCap, 1u << 16, 0);
	if (rc != BT_OK) return rc;
	return worst_status(out, in->n_reads);
}
extern "C" int bt_align_pairs(bt_ctx* c, const bt_read_batch* in1, const bt_read_batch* in2, bt_hit_batch* out, bt_op_counts* counts)
{
	if (!c || !in1 || !in2 || !out) return BT_ERR_ARG;
	if (in1->n_reads == 0) return BT_OK;
	std::lock_guard<std::mutex> lock(g_emu_mutex);
	const int rc = emu_align_pairs(c->ix->emu, &c->pol, in1, in2, out, counts, 0);
	if (rc != BT_OK) return rc;
	return worst_status(out, in1->n_reads);
}
extern "C" void* bt_host_alloc(size_t bytes) { return bytes ? aligned_alloc(256, (bytes + 255u) & ~(size_t)255u) : nullptr; }
extern "C" void bt_host_free(void* p) { free(p); }
/* no device: room for whatever the binary wants in flight (SHIM_STREAM_ROOM: as if the device had room for that many) */
extern "C" int bt_align_stream_room(bt_ctx*, const bt_read_batch*, const bt_hit_batch*, uint32_t* batches)
{
	const char* e = getenv("SHIM_STREAM_ROOM");
	*batches = e ? (uint32_t)atoi(e) : 1000u;
	return BT_OK;
}
/* --stream: the asynchronous entry points, answered synchronously -- a submitted batch is searched at once and is the next
 * one collected; reads come back flagged rather than as an error code, as from the library's stream */
extern "C" int bt_ctx_set_carry(bt_ctx* c, int launches) { return c && launches >= 0 && launches <= 14 ? BT_OK : BT_ERR_ARG; }
extern "C" int bt_align_stream_submit(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, void* tag)
{
	if (!c || !in || !out || !tag || c->best) return BT_ERR_ARG;
	const int rc = bt_align_batch(c, in, out, nullptr);
	if (rc != BT_OK && rc != BT_ERR_OVERFLOW && rc != BT_ERR_READ_SHORT) return rc;
	c->done.push_back(tag);
	return BT_OK;
}
extern "C" int bt_align_stream_tick(bt_ctx* c, uint32_t min_rounds) { (void)min_rounds; return c ? BT_OK : BT_ERR_ARG; }
extern "C" int bt_align_stream_collect(bt_ctx* c, void** tag, int flush)
{
	(void)flush;
	if (!c || !tag) return BT_ERR_ARG;
	*tag = nullptr;
	if (!c->done.empty()) { *tag = c->done.front(); c->done.pop_front(); }
	return BT_OK;
}
