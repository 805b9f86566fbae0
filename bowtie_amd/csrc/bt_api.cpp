/*
 * bt_api.cpp -- the C ABI of include/bowtie_amd.h: index upload, per-GPU context, batch search.
 * Host C++ only; the kernels live in bt_kernels.hip.  No CPU search path exists here: without a
 * HIP device every entry point that computes returns BT_ERR_DEVICE.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <exception>
#include <mutex>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/bowtie_amd.h"
#include "bt_host.h"
#include "bt_kernels.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "bowtie_amd: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	return BT_ERR_DEVICE; } } while (0)

struct bt_index {
	int device = 0;
	int variant = 1;               /* which of .bt2/.ebwt/.bt2l/.ebwtl the base named (bt_host_index_variant) */
	bool has_mirror = false;
	BtIndexHost host[2];           /* big arrays are released after upload; names/plen stay */
	BtIndexDev  dev[2];
	std::vector<void*> allocs;
	uint64_t blk_bytes = 0; uint64_t ebwt_bytes = 0, offs_bytes = 0, jump_bytes = 0;
	std::string base;
	BtRefDev* d_ref = nullptr;     /* the 2-bit reference, loaded on demand (bt_index_load_reference) */
	uint64_t ref_bytes = 0;
	/* the worst-case arenas of the best-first engine's second pass (reads that outgrew their own arena): one set per
	 * index replica = per device, shared by every context on it -- the passes of different contexts take turns on it
	 * through `retryFree` (recorded after a pass, waited for by the next one's stream) */
	/* the locus image of the phase-program engine (bt_rank.h): derived on the device when the first context that can use
	 * it is created (index_ensure_locus), kept for the index's lifetime.  0 not tried, 1 there, -1 not to be had */
	mutable std::mutex locMu;
	mutable int locState = 0;
	mutable uint64_t loc_bytes = 0;
	mutable double loc_build_s = 0;
	mutable std::mutex retryMu;
	mutable uint32_t* retryArenas = nullptr; mutable uint32_t retryArenaLanes = 0;
	mutable hipEvent_t retryFree = nullptr;
};

/* What one call describes to the kernels: the batch (reads in, results out) and the fields of BtHot that follow it */
struct BatchView { BtBatchDev B; const uint8_t* seq; const uint8_t* qual; uint32_t stride, n_reads; };

struct bt_ctx {
	const bt_index* idx = nullptr;
	bt_policy pol;
	BtProgram prog;
	bool best = false;               /* pol.best: the best-first engine (bt_best.h) instead of the phase programs */
	BfProgram bprog;
	BfProgram* d_bprog = nullptr; BtIndexDev* d_ix = nullptr; BtBatchDev* d_batch = nullptr;
	BfProgram* d_bprog_pe = nullptr; bool have_pe = false;      /* the paired program, compiled on first use */
	uint32_t* arenas = nullptr; uint32_t arenaWords = 0; uint32_t arenaLanes = 0; uint32_t arenaAsked = 0;   /* arenaAsked: the lanes wanted when arenaLanes were got */
	uint32_t* retryList = nullptr; uint32_t retryCap = 0;   /* on-device second pass (its arenas: bt_index::retryArenas) */
	hipStream_t stream = nullptr;
	bool own_stream = false;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool timed = false;
	uint32_t nLanes = 0, frCap = 0, entCap = 0, palCap = 0, maxLen = 0;
	uint32_t cus = 0, blocksPerCU = 2;      /* nLanes covers the widest launch (3 blocks per CU) */
	bool rl3 = true;                        /* the three-blocks-per-CU build may be used */
	bool locus = false;                     /* the index has its locus image and this context's launches use it */
	bool jumpOn = true;                     /* ... and its jump table (BT_JUMP=0: A/B, diagnostics) */
	/* descriptors on their way to the device (ctx_h2d): a ring of page-locked slots, so that no asynchronous copy ever
	 * reads memory that has gone out of scope */
	uint8_t* hstage = nullptr; uint32_t hsNext = 0; hipEvent_t hsEv[64] = {}; bool hsUsed[64] = {};
	int occ = 2;
	uint32_t *frames = nullptr, *pairs = nullptr; uint16_t* meta = nullptr; uint64_t* pals = nullptr;
	uint32_t* d_cursor = nullptr;      /* [0] read cursor, [1] mm_pool_used, [7] longest read, [8..10] second pass */
	uint32_t nSlots = 0;
	/* carry-over (bt_kernels.h): the reads the last launch parked and what describes their batch */
	bool carry = false, carryPending = false;
	uint32_t carryAge = 1;                 /* launches a read may be carried through (bt_ctx_set_carry) */
	int carryRl = 0; uint32_t carryBlocks = 0;   /* build and grid of the launches whose lanes the records belong to */
	BtPoolRec* pool = nullptr;             /* [nLanes]: lane g's parked read */
	uint32_t launchSeq = 1;                /* number of the next carry launch; its batch sits in ring[launchSeq % BT_BATCH_RING] */
	uint32_t* d_carry = nullptr;           /* [0..RING) reads parked by the last launch per ring slot, [RING..2 RING) the ring batches' mismatch-pool cursors */
	BatchView ring[BT_BATCH_RING]; bool ringRetry[BT_BATCH_RING] = {}; uint32_t ringMaxLen[BT_BATCH_RING] = {};
	uint32_t* hostParked = nullptr;        /* pinned [16 launches][16 ring slots]: parkedOf after each launch */
	hipEvent_t evLaunch[BT_BATCH_RING] = {};
	BtCold* d_cold_prev = nullptr;         /* the descriptor of a second pass over a ring batch */
	uint32_t* lastMmCursor = nullptr;
	hipEvent_t evSpan = nullptr; bool spanOpen = false; uint32_t spanLaunches = 0;
	hipEvent_t evRing[16][2] = {}; hipEvent_t evFlush[2] = {nullptr, nullptr}; bool flushTimed = false;
	bt_ctx* big = nullptr;             /* lazily created twin with worst-case scratch: reruns reads that overflowed */
	bool is_big = false;
	uint32_t last_retried = 0, last_dev_retried = 0, last_carried = 0;
	uint64_t last_jumps = 0, last_jump_steps = 0;       /* bt_ctx_counts: jump-table look-ups and the reference's steps behind them */
	uint32_t maxLenHint = 0;           /* bt_ctx_set_max_read_len */
	char last_kernel[64] = "";         /* the kernel variant the last batch ran (as rocprofv3 names it) */
	BtCold* d_cold = nullptr;
	BtWarm* d_warm = nullptr;
	unsigned long long* d_counts = nullptr;
	/* staging for the host-pointer entry point */
	void* stage = nullptr; size_t stage_bytes = 0;
	uint32_t last_mm_used = 0;
	uint32_t* iters_dev = nullptr;     /* optional per-read iteration counts (diagnostics) */
	struct bt_stream* hs = nullptr;    /* bt_align_stream_*: staging slots of the batches in flight */
	/* the environment's knobs (diagnostics, A/B, tests), read ONCE per context: a launch path that calls getenv a dozen times
	 * is noise at 22 ms streamed launches (ctx_env) */
	struct { const char* name; uint32_t value; } envc[64]; uint32_t nEnvc = 0;
};

template <class T> static int upload(bt_index* ix, const std::vector<T>& v, const T** out, size_t pad_elems = 0)
{
	void* p = nullptr;
	size_t bytes = (v.size() + pad_elems) * sizeof(T);
	if (bytes == 0) bytes = sizeof(T);
	HIPCHK(hipMalloc(&p, bytes));
	ix->allocs.push_back(p);
	HIPCHK(hipMemset(p, 0, bytes));
	if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
	*out = (const T*)p;
	return BT_OK;
}

extern "C" void bt_policy_default(bt_policy* p)
{
	memset(p, 0, sizeof(*p));
	p->mode = BT_MODE_N; p->mms = 2; p->seed_len = 28; p->qual_thresh = 70; p->max_bts = 125;
	p->maq_round = 1; p->khits = 1; p->mhits = 0xffffffffu;
	p->max_ins = 250; p->mate1_fw = 1; p->pair_tries = 100;
}

extern "C" int bt_has_pe_v1(void)
{
	return 1;
}

extern "C" int bt_index_load(const char* ebwt_base, int need_mirror, int offrate_override, int device,
                             bt_index** out)
{
	if (!ebwt_base || !out) return BT_ERR_ARG;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return BT_ERR_DEVICE;
	HIPCHK(hipSetDevice(device));
	bt_index* ix = new bt_index();
	ix->device = device;
	ix->has_mirror = need_mirror != 0;
	ix->base = ebwt_base;
	ix->variant = bt_host_index_variant(ebwt_base);
	if (ix->variant < 0) { bt_index_free(ix); return BT_ERR_IO; }
	for (int m = 0; m < (need_mirror ? 2 : 1); m++) {
		std::string base = std::string(ebwt_base) + (m ? ".rev" : "");
		int rc;
		try { rc = bt_host_index_load(base, m == 0, offrate_override, &ix->host[m], ix->variant); }
		catch (const std::exception&) { rc = BT_ERR_FORMAT; }          /* a header that asks for absurd array sizes */
		if (rc != BT_OK) { bt_index_free(ix); return rc; }
		BtIndexHost& h = ix->host[m];
		BtIndexDev& d = ix->dev[m];
		bt_host_index_describe(h, &d);
		int r2;
		/* pad the ebwt image by one side pair so that the partner-counter load of the last side
		 * never leaves the allocation; ftab / offs by one 16-byte piece (they are fetched in aligned
		 * 16-byte pieces) */
#if BT_WIDE
		/* the wide build (64-bit rows, bt_rank.h): no side layout -- the loader made the rank blocks and their segment table
		 * on the host (bt_host.cpp: build_blocks); rows numbered from a bias (tests, bt_host.h) shift the row-indexed arrays */
		d.ebwt = nullptr;
		if ((r2 = upload(ix, h.blk, &d.blk, 64)) || (r2 = upload(ix, h.segBase, &d.segBase, 8)) || (r2 = upload(ix, h.ftab, &d.ftab, 4)) ||
		    (r2 = upload(ix, h.eftab, &d.eftab)) || (r2 = upload(ix, h.offs, &d.offs, 4)) ||
		    (r2 = upload(ix, h.rstarts, &d.rstarts)) || (r2 = upload(ix, h.plen, &d.plen))) {
			bt_index_free(ix); return r2;
		}
		bt_host_index_bias(h, &d);
		ix->blk_bytes += h.blk.size();
		std::vector<uint8_t>().swap(h.blk);
#else
		if ((r2 = upload(ix, h.ebwt, &d.ebwt, 128)) || (r2 = upload(ix, h.ftab, &d.ftab, 4)) ||
		    (r2 = upload(ix, h.eftab, &d.eftab)) || (r2 = upload(ix, h.offs, &d.offs, 4)) ||
		    (r2 = upload(ix, h.rstarts, &d.rstarts)) || (r2 = upload(ix, h.plen, &d.plen))) {
			bt_index_free(ix); return r2;
		}
		{
			/* the rank blocks the search kernels query (bt_rank.h), derived on the device from the side layout just
			 * uploaded: 32 bytes per 64 BWT rows */
			const uint64_t nb = bt_blk_count(h.len);
			uint8_t* blk = nullptr;
			if (hipMalloc((void**)&blk, nb * BT_BLK_BYTES) != hipSuccess) { bt_index_free(ix); return BT_ERR_DEVICE; }
			ix->allocs.push_back(blk);
			if (bt_launch_blk_build(&d, blk, (uint32_t)nb, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) { bt_index_free(ix); return BT_ERR_DEVICE; }
			d.blk = blk;
			ix->blk_bytes += nb * BT_BLK_BYTES;
		}
		{
			/* the jump table (bt_rank.h): the range behind a search's first 14 characters, derived from ftab and the rank blocks.
			 * 10 bytes per entry, 2.7 GB per index at 14 characters; only for genomes that can use it (BT_JUMP_CHARS=<n> says
			 * otherwise, 0 = none) */
			uint32_t K = (uint64_t)h.len >= (1ull << 22) ? 14u : 0u;
			if (const char* e = getenv("BT_JUMP_CHARS")) K = (uint32_t)atoi(e);
			if (K > 15u) K = 15u;
			d.jump = nullptr; d.jumpMeta = nullptr; d.jumpChars = 0;
			if (K > (uint32_t)h.ftabChars && h.ftabChars >= 1 && K - (uint32_t)h.ftabChars <= 7u) {
				const uint64_t n = 1ull << (2u * K);
				uint32_t* jt = nullptr; uint16_t* jm = nullptr;
				size_t freeB = 0, totB = 0;
				const bool room = hipMemGetInfo(&freeB, &totB) == hipSuccess && (uint64_t)freeB > n * 10u + (totB / 8u);
				if (room && hipMalloc((void**)&jt, n * 8u) == hipSuccess && hipMalloc((void**)&jm, n * 2u + 16u) == hipSuccess) {
					ix->allocs.push_back(jt); ix->allocs.push_back(jm);
					if (bt_launch_jump_build(&d, K, jt, jm, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) { bt_index_free(ix); return BT_ERR_DEVICE; }
					d.jump = jt; d.jumpMeta = jm; d.jumpChars = K;
					ix->jump_bytes += n * 10u;
				} else {
					(void)hipGetLastError();
					if (jt) (void)hipFree(jt);
					if (getenv("BT_VERBOSE")) fprintf(stderr, "bowtie_amd: no room for the jump table (%.1f GB): searches start from the index's own ftab\n", n * 10.0 / 1e9);
				}
			}
		}
#endif
		ix->ebwt_bytes += h.ebwt.size(); ix->offs_bytes += h.offs.size() * sizeof(bt_row);
		std::vector<uint8_t>().swap(h.ebwt);
		std::vector<bt_row>().swap(h.offs);
		std::vector<bt_row>().swap(h.ftab);
	}
	if (!need_mirror) ix->dev[1] = ix->dev[0];
	/* the uploads above ran on the null stream (fills included, and a fill returns before it has run): nothing of them is
	 * pending when a context -- whose stream does not wait for the null stream -- is created on this index */
	if (hipDeviceSynchronize() != hipSuccess) { bt_index_free(ix); return BT_ERR_DEVICE; }
	*out = ix;
	return BT_OK;
}

extern "C" void bt_index_info_get(const bt_index* idx, bt_index_info* info)
{
	memset(info, 0, sizeof(*info));
	const BtIndexHost& h = idx->host[0];
	info->len = h.len > 0xffffffffull ? 0xffffffffu : (uint32_t)h.len; info->n_pat = h.nPat; info->n_frag = h.nFrag; info->ftab_chars = (uint32_t)h.ftabChars;
	info->off_rate = (uint32_t)h.offRate; info->z_off = (uint32_t)h.zOff;
	info->ebwt_bytes = idx->ebwt_bytes; info->offs_bytes = idx->offs_bytes;
	info->has_mirror = idx->has_mirror ? 1 : 0;
	info->variant = idx->variant | (h.swapped ? BT_INDEX_SWAPPED : 0);
}
extern "C" int bt_rows64(void) { return BT_WIDE ? 1 : 0; }
extern "C" uint64_t bt_index_len64(const bt_index* idx) { return idx ? (uint64_t)idx->host[0].len : 0; }
extern "C" const char* bt_index_refname(const bt_index* idx, uint32_t tidx)
{
	return tidx < idx->host[0].refnames.size() ? idx->host[0].refnames[tidx].c_str() : nullptr;
}
extern "C" uint32_t bt_index_reflen(const bt_index* idx, uint32_t tidx)
{
	return tidx < idx->host[0].plen.size() ? (uint32_t)idx->host[0].plen[tidx] : 0;
}
extern "C" void bt_index_free(bt_index* idx)
{
	if (!idx) return;
	for (void* p : idx->allocs) (void)hipFree(p);
	if (idx->retryFree) { (void)hipEventSynchronize(idx->retryFree); (void)hipEventDestroy(idx->retryFree); }
	if (idx->retryArenas) (void)hipFree(idx->retryArenas);
	delete idx;
}

/* A host structure to device memory, complete when the call returns -- whatever stream reads it next (the contexts' streams
 * are non-blocking: nothing orders them behind the null stream the copy is issued on) */
static hipError_t h2d_now(void* dst, const void* src, size_t bytes)
{
	hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
	return e;
}

static uint32_t env_u32(const char* name, uint32_t dflt)
{
	const char* v = getenv(name);
	return (v && *v) ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

/* env_u32 through the context's cache: the environment is looked at the first time a context asks for a knob */
static uint32_t ctx_env(bt_ctx* c, const char* name, uint32_t dflt)
{
	for (uint32_t i = 0; i < c->nEnvc; i++) if (c->envc[i].name == name || strcmp(c->envc[i].name, name) == 0) return c->envc[i].value;
	const uint32_t v = env_u32(name, dflt);
	if (c->nEnvc < 64u) { c->envc[c->nEnvc].name = name; c->envc[c->nEnvc].value = v; c->nEnvc++; }
	return v;
}

/* A small host structure (a descriptor block, the cursors' initial values) to device memory, on the context's stream.
 * hipMemcpyAsync from ordinary host memory is staged by the runtime -- usually before the call returns, but that is not a
 * promise: under load (several processes on one GPU) the runtime may page-lock the caller's pages and read them when the
 * stream gets there, and every such source in this file used to be a local variable.  Round 5's stress of the overflow
 * second pass (scripts/r5/retry_stress.py, six processes side by side) caught a launch that had seen garbage descriptors:
 * every read of its batch "outgrew" its arenas, and the second pass's mismatch lists came back empty -- the right hits with
 * wrong mismatch entries, round 3's one-off symptom.  The source is now a slot of page-locked memory that
 * belongs to the context and is not written again before the copy that reads it is done. */
#define BT_HSTAGE_SLOTS 64
#define BT_HSTAGE_SLOT_BYTES 16384
static_assert(sizeof(BtCold) <= BT_HSTAGE_SLOT_BYTES && sizeof(BtWarm) <= BT_HSTAGE_SLOT_BYTES && sizeof(BtBatchDev) <= BT_HSTAGE_SLOT_BYTES, "a descriptor fits a staging slot");
static int ctx_h2d(bt_ctx* c, void* dst, const void* src, size_t bytes)
{
	if (bytes > BT_HSTAGE_SLOT_BYTES) return BT_ERR_ARG;
	if (!c->hstage) HIPCHK(hipHostMalloc((void**)&c->hstage, (size_t)BT_HSTAGE_SLOTS * BT_HSTAGE_SLOT_BYTES));
	const uint32_t k = c->hsNext++ % BT_HSTAGE_SLOTS;
	if (!c->hsEv[k]) HIPCHK(hipEventCreateWithFlags(&c->hsEv[k], hipEventDisableTiming));
	if (c->hsUsed[k]) HIPCHK(hipEventSynchronize(c->hsEv[k]));       /* 64 copies ago: done long since */
	uint8_t* slot = c->hstage + (size_t)k * BT_HSTAGE_SLOT_BYTES;
	memcpy(slot, src, bytes);
	HIPCHK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipEventRecord(c->hsEv[k], c->stream));
	c->hsUsed[k] = true;
	return BT_OK;
}
#define BT_H2D(dst, src, bytes) do { const int rc_ = ctx_h2d(c, (dst), (src), (bytes)); if (rc_ != BT_OK) return rc_; } while (0)

/* The locus image (bt_rank.h: dense suffix array + 48 characters of left context per row, the reversed text, the table of
 * walk lengths), derived from what bt_index_load put on the device: 18.25 bytes per base and index -- HBM capacity spent
 * on the search's dependent chains.  Not built when the device has not that much to spare (the search then stays in row
 * space, as in rounds 1-4) or BT_LOCUS=0 says so.  true: the index has it. */
static bool index_ensure_locus(const bt_index* cidx)
{
	std::lock_guard<std::mutex> lk(cidx->locMu);
	if (cidx->locState != 0) return cidx->locState > 0;
	bt_index* idx = const_cast<bt_index*>(cidx);
	idx->locState = -1;
	if (BT_WIDE) return false;                   /* the wide build searches in row space (bt_rank.h, "the row type") */
	if (env_u32("BT_LOCUS", 1) == 0) return false;
	const int nidx = idx->has_mirror ? 2 : 1;
	uint64_t need = 0;
	for (int m = 0; m < nidx; m++)
		need += ((uint64_t)idx->dev[m].len + 1u) * (sizeof(BtU4) + 2u) + bt_rtxt_words(idx->dev[m].len) * 4u;
	size_t freeB = 0, totalB = 0;
	if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return false;
	/* leave room for what a run allocates afterwards (reads, results, scratch arenas): a quarter of the device or what the
	 * image itself takes, whichever is less */
	const uint64_t reserve = std::min<uint64_t>(totalB / 4u, need) + (256ull << 20);
	if ((uint64_t)freeB < need + reserve) {
		if (getenv("BT_VERBOSE")) fprintf(stderr, "bowtie_amd: no room for the locus image (%.1f GB, %.1f GB free): searching in row space\n", need / 1e9, freeB / 1e9);
		return false;
	}
	hipEvent_t e0 = nullptr, e1 = nullptr;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	(void)hipEventRecord(e0, nullptr);
	std::vector<void*> mine;
	bool ok = true;
	for (int m = 0; m < nidx && ok; m++) {
		BtIndexDev& d = idx->dev[m];
		BtU4* loc = nullptr; uint32_t* rtxt = nullptr; uint16_t* walk = nullptr;
		const uint64_t rows = (uint64_t)d.len + 1u, words = bt_rtxt_words(d.len);
		ok = hipMalloc((void**)&loc, rows * sizeof(BtU4)) == hipSuccess;
		if (ok) mine.push_back(loc);
		ok = ok && hipMalloc((void**)&rtxt, words * 4u) == hipSuccess;
		if (ok) mine.push_back(rtxt);
		ok = ok && hipMalloc((void**)&walk, rows * 2u) == hipSuccess;
		if (ok) mine.push_back(walk);
		ok = ok && hipMemset(rtxt, 0, words * 4u) == hipSuccess;
		ok = ok && bt_launch_loc_build(&d, loc, rtxt, walk, nullptr) == 0;
		if (ok) { d.loc = loc; d.rtxt = rtxt + BT_RTXT_PAD_WORDS; d.walk = walk; }
	}
	(void)hipEventRecord(e1, nullptr);
	ok = ok && hipDeviceSynchronize() == hipSuccess;
	float ms = 0;
	if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	if (!ok) {
		(void)hipGetLastError();
		for (void* p : mine) (void)hipFree(p);
		for (int m = 0; m < nidx; m++) { idx->dev[m].loc = nullptr; idx->dev[m].rtxt = nullptr; idx->dev[m].walk = nullptr; }
		return false;
	}
	for (void* p : mine) idx->allocs.push_back(p);
	if (!idx->has_mirror) idx->dev[1] = idx->dev[0];
	idx->loc_bytes = need; idx->loc_build_s = ms / 1e3;
	idx->locState = 1;
	if (getenv("BT_VERBOSE")) fprintf(stderr, "bowtie_amd: locus image %.2f GB, derived in %.2f s\n", need / 1e9, ms / 1e3);
	return true;
}
/* A/B and diagnostics: a context's launches with (1) or without (0) locus mode; 1 has no effect on an index without the image */
static int ctx_flush_carry(bt_ctx* c);
extern "C" int bt_ctx_set_locus(bt_ctx* c, int on)
{
	if (!c) return BT_ERR_ARG;
	/* reads in flight were parked in the mode they were searched in (a lane in locus mode stands on a text offset, not on a
	 * row): they are finished first, and the stream is idle when the descriptors change */
	if (c->carryPending) { const int rc = ctx_flush_carry(c); if (rc != BT_OK) return rc; }
	if (c->stream) HIPCHK(hipStreamSynchronize(c->stream));
	c->locus = on && c->idx->locState > 0;
	if (c->best && c->d_ix) {
		BtIndexDev dv[2] = {c->idx->dev[0], c->idx->dev[1]};
		if (!c->locus) for (int m = 0; m < 2; m++) { dv[m].loc = nullptr; dv[m].rtxt = nullptr; dv[m].walk = nullptr; }
		HIPCHK(h2d_now(c->d_ix, dv, 2 * sizeof(BtIndexDev)));
	}
	if (c->big) c->big->locus = c->locus;
	return BT_OK;
}
extern "C" int bt_ctx_get_locus(const bt_ctx* c) { return c && c->locus ? 1 : 0; }
/* tests: the image's three arrays of one index copied to host buffers (loc: (len+1) x 16 bytes, rtxt: (len+15)/16 words,
 * walk: (len+1) x 2 bytes; NULL skips one) */
extern "C" int bt_index_locus_copy(const bt_index* idx, int mirror, void* loc, void* rtxt, void* walk)
{
	if (!idx || idx->locState <= 0) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(idx->device));
	const BtIndexDev& d = idx->dev[mirror ? 1 : 0];
	const uint64_t rows = (uint64_t)d.len + 1u;
	if (loc) HIPCHK(hipMemcpy(loc, d.loc, rows * sizeof(BtU4), hipMemcpyDeviceToHost));
	if (rtxt) HIPCHK(hipMemcpy(rtxt, d.rtxt, (((uint64_t)d.len + 15u) / 16u) * 4u, hipMemcpyDeviceToHost));
	if (walk) HIPCHK(hipMemcpy(walk, d.walk, rows * 2u, hipMemcpyDeviceToHost));
	return BT_OK;
}
extern "C" uint64_t bt_index_locus_bytes(const bt_index* idx) { return idx ? idx->loc_bytes : 0; }
extern "C" double bt_index_locus_build_seconds(const bt_index* idx) { return idx ? idx->loc_build_s : 0; }

static void ctx_free_scratch(bt_ctx* c)
{
	if (c->frames) (void)hipFree(c->frames);
	if (c->pairs) (void)hipFree(c->pairs);
	if (c->meta) (void)hipFree(c->meta);
	if (c->pals) (void)hipFree(c->pals);
	c->frames = c->pairs = nullptr; c->meta = nullptr; c->pals = nullptr;
}

/* (re)size the per-slot arenas for reads up to maxLen: one slot per resident lane.  With carry-over also the pool
 * of parked-lane records and the counters around it. */
static int ctx_ensure_scratch(bt_ctx* c, uint32_t maxLen, bool carry)
{
	if (carry && !c->pool) {
		HIPCHK(hipMalloc((void**)&c->pool, (size_t)c->nLanes * sizeof(BtPoolRec)));
		HIPCHK(hipMemsetAsync(c->pool, 0, (size_t)c->nLanes * sizeof(BtPoolRec), c->stream));      /* (on the context's stream: see ctx_init) */
		HIPCHK(hipMalloc((void**)&c->d_carry, 2 * BT_BATCH_RING * 4));
		HIPCHK(hipMemsetAsync(c->d_carry, 0, 2 * BT_BATCH_RING * 4, c->stream));
		HIPCHK(hipHostMalloc((void**)&c->hostParked, BT_BATCH_RING * BT_BATCH_RING * 4));
		memset(c->hostParked, 0, BT_BATCH_RING * BT_BATCH_RING * 4);
		for (int i = 0; i < BT_BATCH_RING; i++) HIPCHK(hipEventCreateWithFlags(&c->evLaunch[i], hipEventDisableTiming));
		HIPCHK(hipMalloc((void**)&c->d_cold_prev, sizeof(BtCold)));
	}
	if (c->frames && maxLen <= c->maxLen) return BT_OK;
	ctx_free_scratch(c);
	c->maxLen = maxLen < 64 ? 64 : (maxLen > c->maxLen ? maxLen : c->maxLen);
	c->nSlots = c->nLanes;
	const bool seeded = c->pol.mode == BT_MODE_N;
	/* range-stack entries per slot: every frame may span the whole read.  -v k has k+1 frames;
	 * -n: frames are bounded by -e / min penalty (10) unless the read has Phred<5 bases. */
	uint32_t frames = seeded ? 12u : (uint32_t)c->pol.mms + 2u;
	c->frCap = env_u32("BT_FRAME_CAP", seeded ? 64u : 8u);
	c->entCap = (env_u32("BT_ENTRY_CAP", frames * c->maxLen) + 7u) & ~7u;      /* slot regions stay 16-byte aligned */
	c->palCap = env_u32("BT_PARTIAL_CAP", seeded ? (c->pol.mms >= 3 ? 8192u : 1024u) : 1u);
	if (c->is_big) {
		/* worst case: one frame per query position, frame f spanning the remaining len-f positions;
		 * every (position, character) combination of up to seedMms seed mismatches as a seedling */
		const uint64_t L = c->maxLen, S = seeded ? (uint64_t)c->pol.seed_len : 0, n = seeded ? (uint64_t)c->pol.mms : 0;
		c->frCap = (uint32_t)L + 8u;
		c->entCap = (uint32_t)((L * (L + 3u) / 2u + 64u + 7u) & ~7ull);
		uint64_t pals = 16;
		if (n >= 1) pals += 3 * S;
		if (n >= 2) pals += 9 * S * (S - 1) / 2;
		if (n >= 3) pals += 27 * S * (S - 1) * (S - 2) / 6;
		c->palCap = (uint32_t)(pals > (1u << 22) ? (1u << 22) : pals);
	}
	HIPCHK(hipMalloc((void**)&c->frames, (size_t)c->nSlots * c->frCap * BT_FR_WORDS * 4u));
	HIPCHK(hipMalloc((void**)&c->pairs, (size_t)c->nSlots * c->entCap * 8u * sizeof(bt_row)));
	HIPCHK(hipMalloc((void**)&c->meta, (size_t)c->nSlots * c->entCap * 2u));
	HIPCHK(hipMalloc((void**)&c->pals, (size_t)c->nSlots * c->palCap * 8u));
	if (ctx_env(c, "BT_VERBOSE", 0)) {
		size_t freeB = 0, totB = 0;
		(void)hipMemGetInfo(&freeB, &totB);
		fprintf(stderr, "bowtie_amd: context %p%s: scratch for %u lanes, reads <= %u: %u frames, %u range-stack entries, %u seedlings per lane = %.2f GB; device memory %.1f of %.1f GB free\n",
		        (void*)c, c->is_big ? " (second pass)" : "", c->nSlots, c->maxLen, c->frCap, c->entCap, c->palCap,
		        (double)c->nSlots * ((double)c->frCap * BT_FR_WORDS * 4.0 + (double)c->entCap * (8.0 * sizeof(bt_row) + 2.0) + (double)c->palCap * 8.0) / 1e9, freeB / 1e9, totB / 1e9);
	}
	return BT_OK;
}

static int ctx_init(bt_ctx* c, const bt_index* idx, const bt_policy* pol, void* stream);
static void ctx_free_stream(bt_ctx* c);

extern "C" int bt_ctx_create(const bt_index* idx, const bt_policy* pol, void* stream, bt_ctx** out)
{
	if (!idx || !pol || !out) return BT_ERR_ARG;
	*out = nullptr;
	bt_ctx* c = new bt_ctx();
	const int rc = ctx_init(c, idx, pol, stream);
	if (rc != BT_OK) { bt_ctx_destroy(c); return rc; }      /* whatever was allocated before the failure goes back */
	*out = c;
	return BT_OK;
}

static int ctx_init(bt_ctx* c, const bt_index* idx, const bt_policy* pol, void* stream)
{
	c->idx = idx; c->pol = *pol;
	c->best = pol->best != 0 || pol->pe_v1 != 0;
	c->pol.best = c->best ? 1 : 0;
	int rc = c->best ? bt_host_compile_best(*pol, &c->bprog) : bt_host_compile_program(*pol, &c->prog);
	if (rc != BT_OK) return rc;
	bool need_mirror = c->best && c->bprog.needMirror;
	for (int i = 0; !c->best && i < c->prog.nsteps; i++) need_mirror |= c->prog.steps[i].mirror != 0;
	if (need_mirror && !idx->has_mirror) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(idx->device));
	/* the phase-program engine leaves row space where a range is one row, if the index can have its locus image; the best-first
	 * engine takes a reported row's offset from the image's dense suffix array instead of walking to a sampled row (bt_best.h:
	 * ch_row_set; BT_BEST_LOCUS=0: it walks, as in rounds 2-4) */
	c->locus = index_ensure_locus(idx) && (!c->best || env_u32("BT_BEST_LOCUS", 1) != 0);
	c->jumpOn = env_u32("BT_JUMP", 1) != 0;
	if (stream) c->stream = (hipStream_t)stream;
	else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
	HIPCHK(hipEventCreate(&c->ev0));
	HIPCHK(hipEventCreate(&c->ev1));
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, idx->device));
	c->occ = (int)env_u32("BT_OCC", 2);               /* waves/SIMD the kernel variant is built for */
	if (c->occ < 1) c->occ = 1;
	if (c->occ > 4) c->occ = 4;
	c->cus = (uint32_t)prop.multiProcessorCount;
	c->blocksPerCU = env_u32("BT_BLOCKS_PER_CU", (uint32_t)c->occ);
	/* reads of <= 104 bases run three blocks per CU (the LDS diet of the read-in-LDS build: measured
	 * +15..23 % over two blocks, profiles/README.md); BT_NO_RL3=1 keeps every launch at blocksPerCU */
	c->rl3 = env_u32("BT_NO_RL3", 0) == 0 && c->occ == 2;
	c->nLanes = c->cus * (c->rl3 && c->blocksPerCU < 3u ? 3u : c->blocksPerCU) * BT_BLOCK;
	/* Everything a context's launches read or update is initialised ON THE CONTEXT'S STREAM, never on the null stream: that
	 * stream is non-blocking (it does not wait for the null stream), and hipMemset returns before the fill has run.  Until
	 * round 6 these were null-stream hipMemset calls; with several processes on one GPU such a fill could run late -- after
	 * the context's first launch had started -- and zero the read and mismatch-pool cursors under a running kernel: pool
	 * entries handed out twice, "the right hit with another read's mismatch list" (DESIGN.md 4.3;
	 * tests/test_gpu_parity.py::test_gpu_fresh_context_is_ordered_on_its_own_stream holds the null stream busy to show it). */
	HIPCHK(hipMalloc((void**)&c->d_cursor, 64));
	HIPCHK(hipMemsetAsync(c->d_cursor, 0, 64, c->stream));
	/* carry-over between the launches of this context (bt_kernels.h): asked for with bt_ctx_set_carry or BT_CARRY=1 */
	c->carryAge = env_u32("BT_CARRY", 0);
	if (c->carryAge > BT_BATCH_RING - 2) c->carryAge = BT_BATCH_RING - 2;
	c->carry = c->carryAge != 0;
	HIPCHK(hipEventCreate(&c->evSpan));
	for (int i = 0; i < 16; i++) for (int k = 0; k < 2; k++) HIPCHK(hipEventCreate(&c->evRing[i][k]));
	for (int k = 0; k < 2; k++) HIPCHK(hipEventCreate(&c->evFlush[k]));
	HIPCHK(hipMalloc((void**)&c->d_cold, sizeof(BtCold)));
	HIPCHK(hipMalloc((void**)&c->d_warm, sizeof(BtWarm)));
	HIPCHK(hipMalloc((void**)&c->d_counts, (CN_N + PS_N) * sizeof(unsigned long long)));
	HIPCHK(hipMemsetAsync(c->d_counts, 0, (CN_N + PS_N) * sizeof(unsigned long long), c->stream));
	if (c->best) {
		HIPCHK(hipMalloc((void**)&c->d_bprog, sizeof(BfProgram)));
		HIPCHK(hipMalloc((void**)&c->d_ix, 2 * sizeof(BtIndexDev)));
		HIPCHK(hipMalloc((void**)&c->d_batch, sizeof(BtBatchDev)));
		HIPCHK(h2d_now(c->d_bprog, &c->bprog, sizeof(BfProgram)));
		{
			BtIndexDev dv[2] = {idx->dev[0], idx->dev[1]};
			if (!c->locus) for (int m = 0; m < 2; m++) { dv[m].loc = nullptr; dv[m].rtxt = nullptr; dv[m].walk = nullptr; }
			HIPCHK(h2d_now(c->d_ix, dv, 2 * sizeof(BtIndexDev)));
		}
		/* blocks per CU = waves per SIMD the best-first kernel was compiled for (bt_best_kernels.hip, BT_BEST_MIN_BLOCKS);
		 * BT_BEST_BLOCKS_PER_CU overrides it for A/B runs */
		{
			/* which of the two kernels this context's launches will run (run_best_device): the call-by-call one for small
			 * indexes and for PairedBWAlignerV1, the wavefront automaton otherwise */
			const uint32_t nv = env_u32("BT_BEST_NESTED", 2u);
			const bool small = (uint64_t)idx->dev[0].len < (512ull << 20);
			const bool nested = pol->pe_v1 != 0 || (nv != 2u ? nv != 0u : small);
			c->nLanes = c->cus * env_u32("BT_BEST_BLOCKS_PER_CU", bt_best_blocks_per_cu(nested ? 1 : 0)) * BT_BLOCK;
		}
	}
	return BT_OK;
}

extern "C" void bt_ctx_destroy(bt_ctx* c)
{
	if (!c) return;
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->big) bt_ctx_destroy(c->big);
	ctx_free_scratch(c);
	if (c->d_cursor) (void)hipFree(c->d_cursor);
	if (c->d_cold) (void)hipFree(c->d_cold);
	if (c->d_warm) (void)hipFree(c->d_warm);
	if (c->pool) (void)hipFree(c->pool);
	if (c->d_carry) (void)hipFree(c->d_carry);
	if (c->hostParked) (void)hipHostFree(c->hostParked);
	if (c->hstage) (void)hipHostFree(c->hstage);
	for (int i = 0; i < 64; i++) if (c->hsEv[i]) (void)hipEventDestroy(c->hsEv[i]);
	for (int i = 0; i < BT_BATCH_RING; i++) if (c->evLaunch[i]) (void)hipEventDestroy(c->evLaunch[i]);
	if (c->d_cold_prev) (void)hipFree(c->d_cold_prev);
	if (c->evSpan) (void)hipEventDestroy(c->evSpan);
	ctx_free_stream(c);
	for (int i = 0; i < 16; i++) for (int k = 0; k < 2; k++) if (c->evRing[i][k]) (void)hipEventDestroy(c->evRing[i][k]);
	for (int k = 0; k < 2; k++) if (c->evFlush[k]) (void)hipEventDestroy(c->evFlush[k]);
	if (c->d_counts) (void)hipFree(c->d_counts);
	if (c->d_bprog) (void)hipFree(c->d_bprog);
	if (c->d_bprog_pe) (void)hipFree(c->d_bprog_pe);
	if (c->d_ix) (void)hipFree(c->d_ix);
	if (c->d_batch) (void)hipFree(c->d_batch);
	if (c->arenas) (void)hipFree(c->arenas);
	if (c->retryList) (void)hipFree(c->retryList);
	if (c->stage) (void)hipFree(c->stage);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->own_stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

/* list of the reads a second pass takes: room for one read in 16 (at least 64 K); the few that might not fit keep
 * their BT_STF_OVERFLOW flag */
static int ctx_ensure_retry_list(bt_ctx* c, uint32_t n_reads)
{
	uint32_t want = n_reads / 16u;
	if (want < 65536u) want = n_reads < 65536u ? n_reads : 65536u;
	if (c->retryCap >= want && c->retryList) return BT_OK;
	if (c->retryList) (void)hipFree(c->retryList);
	c->retryList = nullptr; c->retryCap = 0;
	HIPCHK(hipMalloc((void**)&c->retryList, (size_t)want * 4u));
	c->retryCap = want;
	return BT_OK;
}

/* the best-first engine: one launch of bt_best_kernel, every lane with its own arena */
static int run_best_device(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, unsigned long long* counts_dev,
                           const bt_read_batch* in2 = nullptr)
{
	/* arena words per lane: typical reads need a few thousand; a read that outgrows its arena is
	 * flagged (BT_STF_OVERFLOW) and re-run by bt_align_batch through the twin context's 16 MB arenas */
	/* 64 K words (256 KB) per lane: on the hg19-scale index a few per cent of 100-bp -n 2 --best reads need between
	 * 16 K and 64 K words -- reads with hundreds of seed extenders, the slow ones -- and whatever outgrows its arena
	 * lands in the second pass below, which has a thousand lanes instead of a quarter of a million: with 16 K-word
	 * arenas that pass took minutes per million reads (profiles/r3/best_arena.txt) */
	/* (pairs: 32 K words.  Their searches are short -- 50-bp mates, -n 1 -- and half the arena is +4.5 % on config 5's share
	 * (profiles/r6/call7_*: the second pass takes the few more pairs that outgrow it); single-end -n 2 --best loses 18 % to it) */
	const uint32_t words = c->is_big ? (1u << 22) : ctx_env(c, "BT_BEST_ARENA_WORDS", in2 ? 32768u : 65536u);
	/* one arena per lane that the launch can use: a small batch does not fill the grid, and 64 KB x 393 216 lanes
	 * (six blocks per CU) are 26 GB that a thousand-read batch has no use for */
	uint32_t lanes = c->is_big ? (c->nLanes > 256u ? 256u : c->nLanes) : c->nLanes;
	{
		const uint64_t need = ((uint64_t)in->n_reads + BT_BLOCK - 1u) / BT_BLOCK * BT_BLOCK;
		if (need < lanes) lanes = (uint32_t)need;
		if (lanes < BT_BLOCK) lanes = BT_BLOCK;
	}
	if (!c->arenas || c->arenaWords != words || (c->arenaLanes < lanes && c->arenaAsked < lanes)) {
		if (c->arenas) (void)hipFree(c->arenas);
		c->arenas = nullptr; c->arenaLanes = 0; c->arenaAsked = lanes;
		/* a context takes at most 45 % of what the device has free now (BT_BEST_ARENA_FRAC, per cent): the contexts of one
		 * device -- bowtie-amd keeps --inflight of them, and another set for the unpaired records of a --12 file -- then
		 * all fit, the later ones with fewer lanes; and an allocation that fails all the same is tried again with half
		 * the lanes rather than ending the run (a launch with fewer lanes is slower, not wrong) */
		size_t freeB = 0, totB = 0;
		if (hipMemGetInfo(&freeB, &totB) == hipSuccess) {
			/* tests: BT_FAKE_FREE_MB pretends the device has that little free, so that the fewer-lanes path runs on a box
			 * that has plenty */
			const uint32_t fakeMB = ctx_env(c, "BT_FAKE_FREE_MB", 0);
			if (fakeMB && ((size_t)fakeMB << 20) < freeB) freeB = (size_t)fakeMB << 20;
			const uint64_t budget = (uint64_t)freeB / 100u * ctx_env(c, "BT_BEST_ARENA_FRAC", 45u);
			const uint64_t fit = budget / ((uint64_t)words * 4u) / BT_BLOCK * BT_BLOCK;
			if (fit < lanes) lanes = fit < BT_BLOCK ? BT_BLOCK : (uint32_t)fit;
		} else (void)hipGetLastError();
		for (;;) {
			if (hipMalloc((void**)&c->arenas, (size_t)lanes * words * 4u) == hipSuccess) break;
			(void)hipGetLastError();
			c->arenas = nullptr;
			if (lanes <= BT_BLOCK) { fprintf(stderr, "bowtie_amd: no device memory for the best-first arenas (%u words per lane)\n", words); return BT_ERR_DEVICE; }
			lanes = lanes / 2u / BT_BLOCK * BT_BLOCK;
			if (lanes < BT_BLOCK) lanes = BT_BLOCK;
		}
		c->arenaWords = words; c->arenaLanes = lanes;
		if (ctx_env(c, "BT_VERBOSE", 0)) fprintf(stderr, "bowtie_amd: best-first context %p%s: arenas for %u lanes (%u asked) x %u words = %.2f GB; device memory %.1f of %.1f GB free\n",
		                                       (void*)c, c->is_big ? " (second pass)" : "", lanes, c->arenaAsked, words, (double)lanes * words * 4.0 / 1e9, freeB / 1e9, totB / 1e9);
		/* fewer lanes than asked for is slower, not wrong -- but say so once (BT_VERBOSE), and ask again when the caller's
		 * batches still want more and the device may have room by then: arenaAsked stands only as long as memory is short */
		if (lanes < c->arenaAsked) {
			if (ctx_env(c, "BT_VERBOSE", 0)) fprintf(stderr, "bowtie_amd: best-first arenas for %u lanes instead of %u (device memory: %.1f of %.1f GB free)\n",
			                                  lanes, c->arenaAsked, freeB / 1e9, totB / 1e9);
			size_t f2 = 0, t2 = 0;
			if (hipMemGetInfo(&f2, &t2) == hipSuccess && (uint64_t)f2 / 100u * ctx_env(c, "BT_BEST_ARENA_FRAC", 45u) >= (uint64_t)c->arenaAsked * words * 4u)
				c->arenaAsked = lanes;        /* there is room after all (another context went away): the next call may grow */
		}
	}
	if (c->arenaLanes < lanes) lanes = c->arenaLanes;
	BtBatchDev B;
	memset(&B, 0, sizeof(B));
	B.seq = in->seq; B.qual = in->qual; B.len = in->len; B.seed = in->seed;
	B.n_reads = in->n_reads; B.stride = in->stride;
	B.hits = (BtHitRec*)out->hits; B.hit_cap = out->hit_cap; B.n_hits = out->n_hits; B.status = out->status;
	B.mm_pool = out->mm_pool; B.mm_pool_cap = out->mm_pool ? out->mm_pool_cap : 0;
	B.mm_pool_used = c->d_cursor + 1;
	if (in2) { B.seq2 = in2->seq; B.qual2 = in2->qual; B.len2 = in2->len; B.seed2 = in2->seed; B.stride2 = in2->stride; }
	BT_H2D(c->d_batch, &B, sizeof(B));
	const uint32_t init[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	BT_H2D(c->d_cursor, init, sizeof(init));
	BtBestArgs A;
	A.prog = in2 ? c->d_bprog_pe : c->d_bprog; A.ix = c->d_ix; A.batch = c->d_batch;
	A.ref = in2 ? c->idx->d_ref : nullptr;
	A.arenas = c->arenas; A.arenaWords = words; A.nextRead = c->d_cursor;
	A.counts = counts_dev ? counts_dev : c->d_counts;
	uint32_t nBlocks = (in->n_reads + BT_BLOCK - 1) / BT_BLOCK;
	if (nBlocks > lanes / BT_BLOCK) nBlocks = lanes / BT_BLOCK;
	A.workList = nullptr; A.workCount = nullptr; A.workCap = 0; A.laneStride = 1;
	/* Which loop: the wavefront automaton shares the waits of a wavefront's lanes, the call-by-call kernel spends fewer
	 * instructions per lane.  Measured (profiles/r4/sixth_call_automaton_AB.txt): on the hg19-scale index, where every rank
	 * is an HBM latency, the automaton is 1.38x (pairs, BASELINE config 5) and 1.50x (single-end --best) as fast; on
	 * e_coli, whose index sits in the L2 caches, 0.48x / 0.72x.  So: the automaton when the index does not fit the
	 * Infinity Cache (256 MB; the rank blocks are half a byte per base), BT_BEST_NESTED=0/1 to say otherwise. */
	{
		const uint32_t nv = ctx_env(c, "BT_BEST_NESTED", 2u);           /* 0 / 1 say which; unset: by the index's size */
		const bool small = (uint64_t)c->idx->dev[0].len < (512ull << 20);
		A.nested = (in2 && c->pol.pe_v1) || (nv != 2u ? nv != 0u : small) ? 1u : 0u;
	}
	/* the gates' defaults: scripts/best_wave_model.py's pick, then the GPU A/B of profiles/r4/; set again in round 6 for three
	 * blocks per CU with the leaf in LDS (profiles/r6/call6_*, call7_*: a sweep of each gate at hg19 scale).  Pairs want their
	 * cold sweeps rarer -- a pair's cold work (the driver's advance between two leaves, the mate's window scan) is long, and a
	 * sweep for 16 lanes keeps 48 hot ones waiting: 32 is +7 %, 40-48 another 1 % -- single reads do not (-9 %); pairs take
	 * their next read when 24 lanes wait for one (+2.4 %), single reads when 8 do (+4 %); both want streaks' ends gathered
	 * longer (every 8th round or 40 lanes) */
	A.coldMin = ctx_env(c, "BT_BEST_COLD_MIN", in2 ? 40u : 16u); A.takeMin = ctx_env(c, "BT_BEST_TAKE_MIN", in2 ? 24u : 8u);
	A.sendPeriod = ctx_env(c, "BT_BEST_SEND_PERIOD", 8); A.sendMin = ctx_env(c, "BT_BEST_SEND_MIN", 40);
	A.sweepTwice = ctx_env(c, "BT_BEST_SWEEP_TWICE", 0);
	/* per-launch HIP events, as on the phase-program path (bt_ctx_span_ms / bt_ctx_launch_ms) */
	if (!c->spanOpen) { HIPCHK(hipEventRecord(c->evSpan, c->stream)); c->spanOpen = true; c->spanLaunches = 0; c->flushTimed = false; }
	hipEvent_t* ring = c->evRing[c->spanLaunches & 15u];
	c->spanLaunches++;
	HIPCHK(hipEventRecord(ring[0], c->stream));
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	snprintf(c->last_kernel, sizeof(c->last_kernel), A.nested ? "bt_best_nested_kernel" : "bt_best_kernel");
	if (bt_launch_best(&A, nBlocks, c->stream) != 0) return BT_ERR_DEVICE;
	if (!c->is_big && ctx_env(c, "BT_BEST_DEVICE_RETRY", 1)) {
		/* reads that outgrew their arena: collected and searched again on the stream, 1024 lanes with 16 MB
		 * arenas each -- the caller of the device-pointer entry points sees finished results only.  (256 lanes were
		 * tried to save memory: on the hg19-scale index enough reads come here that the pass then takes several
		 * times as long as the main launch, profiles/r3/.) */
		const uint32_t bigWords = 1u << 22;
		uint32_t bigLanes = ctx_env(c, "BT_BEST_RETRY_LANES", in->n_reads >= (1u << 18) ? 1024u : 256u);
		const int rrc = ctx_ensure_retry_list(c, in->n_reads);
		if (rrc != BT_OK) return rrc;
		if (bt_launch_collect_flagged(out->status, in->n_reads, BT_STF_OVERFLOW, c->retryList, c->d_cursor + 2, c->retryCap, c->stream) != 0) return BT_ERR_DEVICE;
		/* the arenas of this pass belong to the index replica: whoever enqueues a pass holds the lock while it does, waits
		 * (on its stream) for the pass before it and leaves the event for the next */
		const bt_index* ix = c->idx;
		std::lock_guard<std::mutex> lock(ix->retryMu);
		if (!ix->retryFree) HIPCHK(hipEventCreateWithFlags(&ix->retryFree, hipEventDisableTiming));
		if (ix->retryArenas && ix->retryArenaLanes < bigLanes) {
			size_t freeB = 0, totB = 0;
			const bool room = hipMemGetInfo(&freeB, &totB) == hipSuccess && (uint64_t)freeB / 2u > (uint64_t)bigLanes * bigWords * 4u;
			if (room) { HIPCHK(hipEventSynchronize(ix->retryFree)); (void)hipFree(ix->retryArenas); ix->retryArenas = nullptr; ix->retryArenaLanes = 0; }
			else bigLanes = ix->retryArenaLanes;               /* no room to grow: the pass runs on what there is */
		}
		while (!ix->retryArenas) {
			if (hipMalloc((void**)&ix->retryArenas, (size_t)bigLanes * bigWords * 4u) == hipSuccess) { ix->retryArenaLanes = bigLanes; break; }
			(void)hipGetLastError();
			ix->retryArenas = nullptr;
			if (bigLanes <= BT_BLOCK) { fprintf(stderr, "bowtie_amd: no device memory for the best-first second pass\n"); return BT_ERR_DEVICE; }
			bigLanes /= 2u;
		}
		if (bigLanes > ix->retryArenaLanes) bigLanes = ix->retryArenaLanes;
		HIPCHK(hipStreamWaitEvent(c->stream, ix->retryFree, 0));
		BtBestArgs A2 = A;
		A2.arenas = ix->retryArenas; A2.arenaWords = bigWords; A2.nextRead = c->d_cursor + 3;
		A2.workList = c->retryList; A2.workCount = c->d_cursor + 2; A2.workCap = c->retryCap;
		/* The second pass lasts as long as its slowest read, and that read is the slower the more of its kind share its
		 * wavefront: the pass's few hundred reads used to fill five wavefronts (a wavefront's 64 lanes take consecutive
		 * items), and rocprofv3 showed the pass to be half of config 5's step.  Only one lane of a wavefront takes reads in
		 * this launch -- the same thousand arenas, a wavefront each.  Measured on config 5's 316 pairs (profiles/r4/
		 * eighth_call_second_pass.txt, ninth_call_second_pass_stride.txt; step = main launch + this pass): 64 to a
		 * wavefront 4.2 s per step (call by call: 5.5 s), every 16th lane 2.8 s (call by call: 2.5 s), one per wavefront
		 * 2.45 s = 10.2 M reads/s against 5.9 M. */
		uint32_t stride = ctx_env(c, "BT_BEST_RETRY_STRIDE", 64);
		if (stride < 1u) stride = 1u;
		if (stride > 64u) stride = 64u;
		A2.laneStride = stride;
		A2.nested = ctx_env(c, "BT_BEST_RETRY_NESTED", 0) ? 1u : A.nested;
		if (bt_launch_best(&A2, bigLanes / BT_BLOCK * stride, c->stream) != 0) return BT_ERR_DEVICE;
		HIPCHK(hipEventRecord(ix->retryFree, c->stream));
	}
	HIPCHK(hipEventRecord(ring[1], c->stream));
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	c->timed = true;
	return BT_OK;
}

/* the twin context that re-runs reads whose search outgrew the per-read scratch: few lanes, worst-case arenas */
static int ctx_ensure_big(bt_ctx* c, uint32_t maxLen, void* stream)
{
	if (c->big) return BT_OK;
	bt_ctx* b = nullptr;
	const int rc = bt_ctx_create(c->idx, &c->pol, stream, &b);
	if (rc != BT_OK) return rc;
	b->is_big = true;
	b->nLanes = BT_BLOCK * (maxLen > 256 ? 1u : 16u);
	b->cus = 1; b->blocksPerCU = b->nLanes / BT_BLOCK; b->rl3 = false;
	c->big = b;
	return BT_OK;
}

static void fill_index_args(const bt_ctx* c, BtKernelArgs* A, BtWarm* warm)
{
	memset(warm, 0, sizeof(*warm));
	for (int m = 0; m < 2; m++) {
		const BtIndexDev& d = c->idx->dev[m];
		A->H.blk[m] = d.blk; A->H.zBlk[m] = d.zBlk; A->H.zPos[m] = d.zPos;
		warm->zOff[m] = d.zOff; warm->offMask[m] = d.offMask; warm->ftab[m] = d.ftab; warm->offs[m] = d.offs;
		warm->offRate[m] = d.offRate; warm->ftabChars[m] = d.ftabChars; warm->len[m] = d.len;
		for (int k = 0; k < 5; k++) A->H.fchr[m][k] = d.fchr[k];
		if (c->locus) { warm->loc[m] = d.loc; warm->rtxt[m] = d.rtxt; warm->walk[m] = d.walk; }
		if (c->jumpOn) { warm->jump[m] = d.jump; warm->jumpMeta[m] = d.jumpMeta; warm->jumpChars[m] = d.jump ? d.jumpChars : 0u; }
#if BT_WIDE
		A->H.segBase[m] = d.segBase; A->H.segShift = d.segShift; warm->rowLim[m] = d.rowLim;
#endif
	}
	warm->locOn = c->locus ? 1u : 0u;
}

/* Second pass, on the stream, over the reads of `v` whose search outgrew the per-read scratch: collected from the
 * status array and searched again through the twin context's worst-case arenas.  `d_cold` must describe `v` as its
 * current batch (B, curBid and ring[curBid]). */
static int enqueue_retry(bt_ctx* c, const BtKernelArgs& A0, const BatchView& v, const BtCold* d_cold, uint32_t maxLen)
{
	const bt_ctx* b = c->big;
	HIPCHK(hipMemsetAsync(c->d_cursor + 8, 0, 12, c->stream));
	if (bt_launch_collect_flagged(v.B.status, v.n_reads, BT_STF_OVERFLOW, c->retryList, c->d_cursor + 8, c->retryCap, c->stream) != 0)
		return BT_ERR_DEVICE;
	BtKernelArgs R = A0;                              /* same index */
	R.H.seq = v.seq; R.H.qual = v.qual; R.H.stride = v.stride; R.H.n_reads = v.n_reads;
	R.cold = d_cold;
	R.gate = nullptr;
	R.frames = b->frames; R.pairs = b->pairs; R.meta = b->meta; R.pals = b->pals;
	R.nLanes = b->nLanes; R.nSlots = b->nSlots; R.frCap = b->frCap; R.entCap = b->entCap; R.palCap = b->palCap;
	R.nextRead = c->d_cursor + 9;
	R.order = c->retryList; R.orderCount = c->d_cursor + 8; R.orderCap = c->retryCap;
	R.pool = nullptr; R.adopt = 0; R.park = 0; R.parkedOf = nullptr;
	if (bt_launch_search(&R, b->nLanes / BT_BLOCK, c->occ, (maxLen <= BT_RL_MAXLEN && !ctx_env(c, "BT_NO_RL", 0) ? 1 : 0) | (ctx_env(c, "BT_FORCE_EXT", 0) ? BT_RL_FORCE_EXT : 0), c->stream) != 0)
		return BT_ERR_DEVICE;
	return BT_OK;
}

static void fill_cold(const bt_ctx* c, BtCold* cold, const BtBatchDev& cur, uint32_t curBid)
{
	memset(cold, 0, sizeof(*cold));
	cold->P = c->prog;
	cold->ix[0] = c->idx->dev[0]; cold->ix[1] = c->idx->dev[1];
	cold->B = cur; cold->curBid = curBid;
	for (int i = 0; i < BT_BATCH_RING; i++) cold->ring[i] = c->ring[i].B;
	cold->ring[curBid] = cur;
}

/* Finish the reads the last launch parked (carry-over): a launch of the same grid with no fresh reads, in which
 * every lane picks its parked read up and runs it to the end; then the second pass over the batches that are due
 * one.  After it the context holds nothing in flight. */
static int ctx_flush_carry(bt_ctx* c)
{
	if (!c->carryPending) return BT_OK;
	HIPCHK(hipSetDevice(c->idx->device));
	BtKernelArgs A;
	memset(&A, 0, sizeof(A));
	BtWarm warm;
	fill_index_args(c, &A, &warm);
	const uint32_t bid = c->launchSeq & (BT_BATCH_RING - 1u);
	BtCold cold;
	BtBatchDev none;
	memset(&none, 0, sizeof(none));
	fill_cold(c, &cold, none, bid);
	const BatchView& lastB = c->ring[(c->launchSeq - 1u) & (BT_BATCH_RING - 1u)];
	const uint32_t keep = ctx_env(c, "BT_FLUSH_KEEP_BATCH", 0);   /* diagnostics: 1 = the last batch stands in as the current one */
	if (keep) cold.B = lastB.B;
	BT_H2D(c->d_cold, &cold, sizeof(cold));
	BT_H2D(c->d_warm, &warm, sizeof(warm));
	const uint32_t init[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	BT_H2D(c->d_cursor, init, sizeof(init));
	HIPCHK(hipMemsetAsync(c->d_carry, 0, BT_BATCH_RING * 4, c->stream));
	A.H.seq = nullptr; A.H.qual = nullptr; A.H.stride = 0; A.H.n_reads = 0;
	if (keep) { A.H.seq = lastB.seq; A.H.qual = lastB.qual; A.H.stride = lastB.stride; }
	A.cold = c->d_cold; A.warm = c->d_warm;
	A.frames = c->frames; A.pairs = c->pairs; A.meta = c->meta; A.pals = c->pals;
	A.nLanes = c->nLanes; A.nSlots = c->nSlots; A.frCap = c->frCap; A.entCap = c->entCap; A.palCap = c->palCap;
	A.counts = c->d_counts;
	A.nextRead = c->d_cursor;
	A.pool = c->pool; A.launchSeq = c->launchSeq; A.adopt = 1; A.park = 0; A.maxAge = 0; A.parkedOf = c->d_carry;
	HIPCHK(hipEventRecord(c->evFlush[0], c->stream));
	const bool dbg = ctx_env(c, "BT_CARRY_DEBUG", 0) != 0;       /* diagnostics: name and fence every carry launch */
	if (dbg) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[carry] flush launch seq=%u blocks=%u rl=%d ...\n", c->launchSeq, c->carryBlocks, c->carryRl); }
	if (bt_launch_search(&A, c->carryBlocks, c->occ, c->carryRl | (ctx_env(c, "BT_FORCE_EXT", 0) ? BT_RL_FORCE_EXT : 0), c->stream) != 0) return BT_ERR_DEVICE;
	if (dbg) { const int e = (int)hipStreamSynchronize(c->stream); fprintf(stderr, "[carry] flush launch done rc=%d\n", e); }
	c->launchSeq++;
	c->carryPending = false;
	for (int i = 0; i < BT_BATCH_RING; i++) if (c->ringRetry[i]) {
		BtCold pc;
		fill_cold(c, &pc, c->ring[i].B, (uint32_t)i);
		BT_H2D(c->d_cold_prev, &pc, sizeof(pc));
		const int rc = enqueue_retry(c, A, c->ring[i], c->d_cold_prev, c->ringMaxLen[i]);
		if (rc != BT_OK) return rc;
		c->ringRetry[i] = false;
	}
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipEventRecord(c->evFlush[1], c->stream));
	c->flushTimed = true;
	return BT_OK;
}

/* lens_on_device: maxLen is only the row stride (the lengths are in HBM); async: the caller does not wait for this
 * batch before handing over the next -- carry-over and the on-stream second pass apply */
static int run_device(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, uint32_t maxLen,
                      unsigned long long* counts_dev, bool lens_on_device, bool async, bool retry_on_stream = true,
                      uint32_t* mmCursorDev = nullptr)
{
	if (in->n_reads == 0) { c->timed = false; return BT_OK; }
	if (!in->seq || !in->qual || !in->len || !in->seed || !out->hits || !out->n_hits || !out->status ||
	    out->hit_cap == 0 || in->stride == 0 || (in->stride & 15u) != 0 ||
	    ((uintptr_t)in->seq & 15u) != 0 || ((uintptr_t)in->qual & 15u) != 0) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	if (c->best) return run_best_device(c, in, out, counts_dev);
	int rc;
	BtKernelArgs A;
	memset(&A, 0, sizeof(A));
	/* short reads (all of today's sequencers' single-end lengths up to 112) keep the whole read in LDS */
	int rl = (maxLen <= BT_RL_MAXLEN && !ctx_env(c, "BT_NO_RL", 0)) ? 1 : 0;
	bool both = false;                /* longest read known on the device only: enqueue both builds, gated */
	if (rl && c->rl3) {
		/* only the row stride is known here: one small reduction over len[] settles it, on the stream (below) */
		if (lens_on_device && maxLen > BT_RL3_MAXLEN) both = true;
		if (maxLen <= BT_RL3_MAXLEN || both) rl = 2;          /* three blocks per CU: the LDS diet */
	}
	/* carry-over (bt_kernels.h): batches the caller does not wait for, on a context that asked for it, reads in LDS,
	 * one build; every such launch has the same grid, because a parked read belongs to its lane */
	const bool carry = c->carry && async && !c->is_big && rl != 0 && !both && counts_dev == nullptr;
	uint32_t gridBlocks = c->cus * (rl == 2 ? 3u : c->blocksPerCU);
	{
		const uint32_t lim = ctx_env(c, "BT_MAX_BLOCKS", 0);   /* tests: a small grid makes small batches drain */
		if (lim && lim < gridBlocks) gridBlocks = lim;
	}
	if (c->carryPending && (!carry || rl != c->carryRl || gridBlocks != c->carryBlocks || maxLen > c->maxLen)) {
		if ((rc = ctx_flush_carry(c)) != BT_OK) return rc;
	}
	if ((rc = ctx_ensure_scratch(c, maxLen, carry)) != BT_OK) return rc;
	/* reads that outgrow their scratch can be searched again on the stream (below), through the twin context's
	 * worst-case arenas, so that the device-pointer entry point hands back finished results.
	 * Off by default (BT_DEVICE_RETRY=1 turns it on): the second pass runs the EXT instances of the kernel, whose fault on
	 * two inputs of the simple_tests suite was fixed too late in round 2 for the whole GPU suite to run through them
	 * (DESIGN.md 4.4). */
	const bool devRetry = async && retry_on_stream && !c->is_big && ctx_env(c, "BT_DEVICE_RETRY", 1);
	if (devRetry) {
		if ((rc = ctx_ensure_big(c, maxLen, c->stream)) != BT_OK) return rc;
		if ((rc = ctx_ensure_scratch(c->big, maxLen, false)) != BT_OK) return rc;
		if ((rc = ctx_ensure_retry_list(c, in->n_reads)) != BT_OK) return rc;
	}
	const uint32_t bid = carry ? (c->launchSeq & (BT_BATCH_RING - 1u)) : 0u;
	BatchView cur;
	memset(&cur, 0, sizeof(cur));
	cur.B.seq = in->seq; cur.B.qual = in->qual; cur.B.len = in->len; cur.B.seed = in->seed;
	cur.B.n_reads = in->n_reads; cur.B.stride = in->stride;
	cur.B.hits = (BtHitRec*)out->hits; cur.B.hit_cap = out->hit_cap;
	cur.B.n_hits = out->n_hits; cur.B.status = out->status;
	cur.B.mm_pool = out->mm_pool; cur.B.mm_pool_cap = out->mm_pool ? out->mm_pool_cap : 0;
	/* the batch's mismatch-pool cursor: the caller's own word (a stream slot's: it outlives the launches that follow), or
	 * the ring's with carry-over, or the context's */
	cur.B.mm_pool_used = mmCursorDev ? mmCursorDev : (carry ? c->d_carry + BT_BATCH_RING + bid : c->d_cursor + 1);
	cur.B.iters = c->iters_dev;
	cur.seq = in->seq; cur.qual = in->qual; cur.stride = in->stride; cur.n_reads = in->n_reads;
	const bool adopt = carry && c->carryPending;
	if (carry) { c->ring[bid] = cur; c->ringRetry[bid] = devRetry; c->ringMaxLen[bid] = maxLen; }
	BtCold cold;
	fill_cold(c, &cold, cur.B, bid);
	BT_H2D(c->d_cold, &cold, sizeof(cold));
	BtWarm warm;
	fill_index_args(c, &A, &warm);
	A.H.seq = in->seq; A.H.qual = in->qual; A.H.stride = in->stride; A.H.n_reads = in->n_reads;
	BT_H2D(c->d_warm, &warm, sizeof(warm));
	A.cold = c->d_cold; A.warm = c->d_warm;
	A.frames = c->frames; A.pairs = c->pairs; A.meta = c->meta; A.pals = c->pals;
	A.nLanes = c->nLanes; A.nSlots = c->nSlots; A.frCap = c->frCap; A.entCap = c->entCap; A.palCap = c->palCap;
	A.nextRead = c->d_cursor;
	A.counts = counts_dev ? counts_dev : c->d_counts;
	/* [0] read cursor, [1] mismatch-pool cursor, [7] longest read, [8] reads to search again, [9] their cursor */
	const uint32_t init[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	BT_H2D(c->d_cursor, init, sizeof(init));
	if (carry) {
		HIPCHK(hipMemsetAsync(c->d_carry, 0, BT_BATCH_RING * 4, c->stream));                 /* this launch's parked counts */
		if (!mmCursorDev) HIPCHK(hipMemsetAsync(c->d_carry + BT_BATCH_RING + bid, 0, 4, c->stream));   /* this batch's mismatch-pool cursor */
		A.pool = c->pool; A.launchSeq = c->launchSeq; A.adopt = adopt ? 1u : 0u; A.park = 1u;
		A.maxAge = c->carryAge; A.parkedOf = c->d_carry;
		A.parkMinRounds = ctx_env(c, "BT_PARK_MIN_ROUNDS", 0);
	}
	if (mmCursorDev) HIPCHK(hipMemsetAsync(mmCursorDev, 0, 4, c->stream));
	if (both && bt_launch_maxlen(in->len, in->n_reads, c->d_cursor + 7, c->stream) != 0) return BT_ERR_DEVICE;
	if (!c->spanOpen) { HIPCHK(hipEventRecord(c->evSpan, c->stream)); c->spanOpen = true; c->spanLaunches = 0; c->flushTimed = false; }
	hipEvent_t* ring = c->evRing[c->spanLaunches & 15u];
	c->spanLaunches++;
	HIPCHK(hipEventRecord(ring[0], c->stream));
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	A.order = nullptr;
#ifdef BT_TRACE
	uint32_t* d_trace = nullptr;
	const uint32_t traceCap = 4096;
	if (getenv("BT_TRACE_READ")) {
		HIPCHK(hipMalloc((void**)&d_trace, (4 + 12 * traceCap) * 4));
		HIPCHK(hipMemset(d_trace, 0, (4 + 12 * traceCap) * 4));
		A.trace = d_trace; A.traceRead = (uint32_t)atoi(getenv("BT_TRACE_READ")); A.traceCap = traceCap;
	}
#endif
	auto launch_main = [&](int rlv) -> int {
		uint32_t maxBlocks = c->cus * (rlv == 2 ? 3u : c->blocksPerCU);
		const uint32_t lim = ctx_env(c, "BT_MAX_BLOCKS", 0);
		if (lim && lim < maxBlocks) maxBlocks = lim;
		uint32_t nBlocks = (in->n_reads + BT_BLOCK - 1) / BT_BLOCK;
		if (carry || nBlocks > maxBlocks) nBlocks = maxBlocks;     /* carry-over: always the whole grid */
		{
			/* the template instance bt_launch_search picks (bt_kernels.hip) */
			const bool ext = A.pool || A.order || ctx_env(c, "BT_FORCE_EXT", 0);
			const int o = rlv == 2 ? 3 : (rlv ? (c->occ == 1 ? 1 : 2) : (c->occ < 1 ? 1 : (c->occ > 4 ? 4 : c->occ)));
			snprintf(c->last_kernel, sizeof(c->last_kernel), "bt_search_kernel<%d,%s,%s,%s>", o, ext ? "true" : "false",
			         rlv ? "true" : "false", rlv == 2 ? "true" : "false");
		}
		return bt_launch_search(&A, nBlocks, c->occ, rlv | (ctx_env(c, "BT_FORCE_EXT", 0) ? BT_RL_FORCE_EXT : 0), c->stream) != 0 ? BT_ERR_DEVICE : BT_OK;
	};
	if (both) {
		/* the batch's longest read is on the device only (c->d_cursor[7], reduced above): both builds are
		 * enqueued, each gated on it; the one it rules out returns at once.  No host wait. */
		A.gate = c->d_cursor + 7;
		A.gateLo = BT_RL3_MAXLEN + 1u; A.gateHi = 0xffffffffu;
		if ((rc = launch_main(1)) != BT_OK) return rc;
		A.gateLo = 0; A.gateHi = BT_RL3_MAXLEN;
		if ((rc = launch_main(2)) != BT_OK) return rc;
		snprintf(c->last_kernel, sizeof(c->last_kernel), "bt_search_kernel<3|2,*,true,*> (gated)");
	} else {
		A.gate = nullptr;
		const bool dbg = carry && ctx_env(c, "BT_CARRY_DEBUG", 0) != 0;
		if (dbg) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[carry] main launch seq=%u bid=%u adopt=%d n_reads=%u blocks=%u rl=%d maxAge=%u ...\n", c->launchSeq, bid, (int)adopt, in->n_reads, gridBlocks, rl, c->carryAge); }
		if ((rc = launch_main(rl)) != BT_OK) return rc;
		if (dbg) {
			const int e = (int)hipStreamSynchronize(c->stream); fprintf(stderr, "[carry] main launch done rc=%d\n", e);
			/* what the first lanes parked: the automaton state a continuation starts from */
			const uint32_t nd = in->n_reads < 4u ? in->n_reads : 4u;
			std::vector<BtPoolRec> recs(nd);
			if (hipMemcpy(recs.data(), c->pool, nd * sizeof(BtPoolRec), hipMemcpyDeviceToHost) == hipSuccess)
				for (uint32_t g = 0; g < nd; g++) {
					BtLane L; memcpy(&L, recs[g].w, sizeof(L));
					fprintf(stderr, "[carry] pool[%u] stamp=%u%s rd=%u bid=%u state=%u step=%u kind=%u mirror=%u readFw=%u rev=%u qlen=%u plen=%u sd=%u depth=%u d=%u top=%u bot=%u iters=%u nhits=%u stored=%u status=%u req{kind=%u n=%u a=%llx x=%llx}\n",
					        g, recs[g].w[BT_POOL_STAMP_WORD], recs[g].w[BT_POOL_STAMP_WORD] == c->launchSeq ? "(live)" : "(stale)", L.rd, (unsigned)L.bid, (unsigned)L.state, (unsigned)L.step, (unsigned)L.kind,
					        (unsigned)L.mirror, (unsigned)L.readFw, (unsigned)L.rev, (unsigned)L.qlen, (unsigned)L.plen, (unsigned)L.sd, (unsigned)L.depth, (unsigned)L.d, (unsigned)L.top, (unsigned)L.bot,
					        L.iters, L.nhits, (unsigned)L.stored, (unsigned)L.status, recs[g].w[4 * BT_POOL_REQ], recs[g].w[4 * BT_POOL_REQ + 1],
					        (unsigned long long)(((uint64_t)recs[g].w[4 * BT_POOL_REQ + 5] << 32) | recs[g].w[4 * BT_POOL_REQ + 4]), (unsigned long long)(((uint64_t)recs[g].w[4 * BT_POOL_REQ + 7] << 32) | recs[g].w[4 * BT_POOL_REQ + 6]));
				}
		}
	}
	if (carry) {
		/* which ring batches still have reads parked after this launch: to the host, for whoever wants to know when a
		 * batch is complete (bt_align_stream_collect) */
		const uint32_t k = c->launchSeq & (BT_BATCH_RING - 1u);
		HIPCHK(hipMemcpyAsync(c->hostParked + (size_t)k * BT_BATCH_RING, c->d_carry, BT_BATCH_RING * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipEventRecord(c->evLaunch[k], c->stream));
		c->carryPending = true; c->carryRl = rl; c->carryBlocks = gridBlocks;
		c->launchSeq++;
		c->lastMmCursor = mmCursorDev ? mmCursorDev : c->d_carry + BT_BATCH_RING + bid;
		/* the second pass waits for the batch to be complete: ctx_flush_carry */
	} else {
		if (devRetry && (rc = enqueue_retry(c, A, cur, c->d_cold, maxLen)) != BT_OK) return rc;
		c->lastMmCursor = mmCursorDev ? mmCursorDev : c->d_cursor + 1;
	}
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipEventRecord(ring[1], c->stream));
	c->timed = true;
#ifdef BT_TRACE
	if (d_trace) {
		/* copy what was written even if the kernel faulted later: the copy may fail, then nothing is printed */
		std::vector<uint32_t> h(4 + 12 * traceCap);
		(void)hipStreamSynchronize(c->stream);
		if (hipMemcpy(h.data(), d_trace, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
			const uint32_t n = h[0] < traceCap ? h[0] : traceCap;
			fprintf(stderr, "[trace] read %u: %u rounds recorded, kernel %s\n", A.traceRead, h[0], c->last_kernel);
			for (uint32_t k = 0; k < n; k++) {
				const uint32_t* t = h.data() + 4 + 12 * k;
				fprintf(stderr, "[trace] r=%u st=%u step=%u mir=%u fw=%u rev=%u req=%u n=%u a=%08x%08x x=%08x%08x top=%u bot=%u d=%u sd=%u\n",
				        t[0], t[1], t[2] & 255u, (t[2] >> 8) & 1u, (t[2] >> 9) & 1u, (t[2] >> 10) & 1u, t[3], t[4], t[6], t[5], t[8], t[7], t[9], t[10], t[11] & 0xffffu, t[11] >> 16);
			}
		}
		(void)hipFree(d_trace);
	}
#endif
	return BT_OK;
}

extern "C" int bt_align_batch_device(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out,
                                     bt_op_counts* counts_dev)
{
	if (!c || !in || !out) return BT_ERR_ARG;
	/* (a caller's device-side counter block: the kernels tally more words than bt_op_counts has fields -- section timers, the
	 * locus-mode and jump-table tallies that bt_ctx_counts folds into the reference's op counts -- so since 0.2.0 the counters
	 * are read with bt_ctx_counts after bt_ctx_sync and this must be NULL) */
	if (counts_dev) return BT_ERR_UNSUPPORTED;
	/* lengths live in HBM: size the scratch for the row stride (>= every length), or for what the caller vouched for */
	const uint32_t hint = c->maxLenHint && c->maxLenHint < in->stride ? c->maxLenHint : 0u;
	return run_device(c, in, out, hint ? hint : in->stride, (unsigned long long*)counts_dev, hint == 0, true);
}

/* Replaces: BitPairReference's constructor (reference.h:35-240; ebwt_search.cpp:3162-3171 loads it
 * for paired-end runs): <base>.3.ebwt / .4.ebwt into HBM, 2 bits per base plus an N mask. */
extern "C" int bt_index_load_reference(bt_index* ix)
{
	if (!ix) return BT_ERR_ARG;
	if (ix->d_ref) return BT_OK;
	HIPCHK(hipSetDevice(ix->device));
	BtRefHost R;
	int rc;
	try { rc = bt_host_ref_load(ix->base, ix->host[0], &R, ix->variant); }
	catch (const std::exception&) { rc = BT_ERR_FORMAT; }
	if (rc != BT_OK) return rc;
	BtRefDev d;
	memset(&d, 0, sizeof(d));
	int r2;
	/* + 8 words each: the mate finder reads the 2-bit reference and its N mask in 64-bit pieces, three / two at a time */
	if ((r2 = upload(ix, R.bits, &d.bits, 8)) || (r2 = upload(ix, R.nmask, &d.nmask, 8)) || (r2 = upload(ix, R.start, &d.start)) ||
	    (r2 = upload(ix, R.approxLen, &d.approxLen))) return r2;
	d.nRefs = (uint32_t)R.start.size();
	void* p = nullptr;
	HIPCHK(hipMalloc(&p, sizeof(d)));
	ix->allocs.push_back(p);
	HIPCHK(h2d_now(p, &d, sizeof(d)));
	ix->d_ref = (BtRefDev*)p;
	ix->ref_bytes = (R.bits.size() + R.nmask.size()) * 4ull;
	return BT_OK;
}

static int ctx_ensure_paired(bt_ctx* c)
{
	if (!c->best) return BT_ERR_ARG;                 /* PairedBWAlignerV2 is the --best aligner */
	if (c->have_pe) return BT_OK;
	if (!c->idx->d_ref) return BT_ERR_ARG;           /* bt_index_load_reference first */
	BfProgram P;
	int rc = bt_host_compile_best_paired(c->pol, &P);
	if (rc != BT_OK) return rc;
	if (P.needMirror && !c->idx->has_mirror) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	HIPCHK(hipMalloc((void**)&c->d_bprog_pe, sizeof(BfProgram)));
	HIPCHK(h2d_now(c->d_bprog_pe, &P, sizeof(P)));
	c->have_pe = true;
	return BT_OK;
}

extern "C" int bt_align_pairs_device(bt_ctx* c, const bt_read_batch* in1, const bt_read_batch* in2, bt_hit_batch* out,
                                     bt_op_counts* counts_dev)
{
	if (!c || !in1 || !in2 || !out || in1->n_reads != in2->n_reads) return BT_ERR_ARG;
	if (counts_dev) return BT_ERR_UNSUPPORTED;          /* (see bt_align_batch_device) */
	int rc = ctx_ensure_paired(c);
	if (rc != BT_OK) return rc;
	if (in1->n_reads == 0) { c->timed = false; return BT_OK; }
	if (!in1->seq || !in1->qual || !in1->len || !in1->seed || !in2->seq || !in2->qual || !in2->len || !in2->seed ||
	    !out->hits || !out->n_hits || !out->status || out->hit_cap < 2 || (out->hit_cap & 1u) ||
	    (in1->stride & 15u) || (in2->stride & 15u)) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	return run_best_device(c, in1, out, (unsigned long long*)counts_dev, in2);
}


/* What a batch's hit records and its mismatch-pool cursor must agree on, checked on the host before results are handed
 * back: every stored hit's list lies inside the entries the kernel handed out, and (exact != 0: one pass, no read flagged
 * BT_STF_MMPOOL) the lists add up to the cursor -- each entry handed out belongs to exactly one stored hit.  Small batches
 * (and any batch under BT_CHECK=1) are also checked for overlap: the lists, put in order, tile [0, cursor).  A batch that
 * fails comes back as BT_ERR_DEVICE with both numbers, not as plausible results (round 5's driver run: "the right hit
 * with another read's mismatch list", status OK). */
static int check_mm_pool(const bt_hit_batch* out, uint32_t n, uint32_t cursor, bool exact, const char* what)
{
	const uint32_t used = cursor < out->mm_pool_cap ? cursor : out->mm_pool_cap;
	uint64_t sum = 0, outside = 0; bool mmpool = false;
	const size_t slots = (size_t)n * out->hit_cap;
	for (uint32_t i = 0; i < n; i++) mmpool |= (out->status[i] & BT_STF_MMPOOL) != 0;
	auto scan = [&](size_t lo, size_t hi, uint64_t* s, uint64_t* o) {
		uint64_t ss = 0, oo = 0;
		for (size_t k = lo; k < hi; k++) {
			const bt_hit& h = out->hits[k];
			if (!h.nmm) continue;
			ss += h.nmm;
			if ((uint64_t)h.mm_off + h.nmm > used) oo++;
		}
		*s = ss; *o = oo;
	};
	if (slots >= ((size_t)1u << 22)) {
		/* a batch of millions of reads: the pass over its hit records is on the thread that feeds the GPU */
		constexpr int T = 8;
		uint64_t ps[T], po[T];
		std::thread th[T];
		for (int t = 0; t < T; t++) th[t] = std::thread(scan, slots * (size_t)t / T, slots * (size_t)(t + 1) / T, &ps[t], &po[t]);
		for (int t = 0; t < T; t++) { th[t].join(); sum += ps[t]; outside += po[t]; }
	} else scan(0, slots, &sum, &outside);
	bool bad = outside != 0 || (exact && !mmpool && sum != cursor) || (!exact && sum > cursor);
	uint64_t overlaps = 0;
	static const bool checkAll = getenv("BT_CHECK") && atoi(getenv("BT_CHECK")) != 0;
	if (!bad && (slots <= (1u << 20) || checkAll)) {
		std::vector<uint64_t> r;
		for (size_t k = 0; k < slots; k++) if (out->hits[k].nmm) r.push_back(((uint64_t)out->hits[k].mm_off << 16) | out->hits[k].nmm);
		std::sort(r.begin(), r.end());
		uint64_t end = 0;
		for (uint64_t v : r) { if ((v >> 16) < end) overlaps++; end = (v >> 16) + (v & 0xffffu); }
		bad = overlaps != 0;
	}
	if (!bad) return BT_OK;
	fprintf(stderr, "bowtie_amd: %s: the hit records and the mismatch-pool cursor disagree: %llu entries in stored hits, cursor %u (capacity %u), "
	                "%llu lists outside the entries handed out, %llu overlapping -- results withheld\n",
	        what, (unsigned long long)sum, cursor, out->mm_pool_cap, (unsigned long long)outside, (unsigned long long)overlaps);
	return BT_ERR_DEVICE;
}

extern "C" int bt_align_pairs(bt_ctx* c, const bt_read_batch* in1, const bt_read_batch* in2, bt_hit_batch* out,
                              bt_op_counts* counts)
{
	if (!c || !in1 || !in2 || !out || in1->n_reads != in2->n_reads) return BT_ERR_ARG;
	int rc = ctx_ensure_paired(c);
	if (rc != BT_OK) return rc;
	const uint32_t n = in1->n_reads;
	out->mm_pool_used = 0;
	if (n == 0) return BT_OK;
	if (!out->hits || !out->n_hits || !out->status || out->hit_cap < 2 || (out->hit_cap & 1u)) return BT_ERR_ARG;
	for (int m = 0; m < 2; m++) {
		const bt_read_batch* in = m ? in2 : in1;
		if (!in->seq || !in->qual || !in->len || !in->seed || (in->stride & 15u) != 0) return BT_ERR_ARG;
		for (uint32_t i = 0; i < n; i++) if (in->len[i] > 1024 || in->len[i] > in->stride) return BT_ERR_ARG;
	}
	HIPCHK(hipSetDevice(c->idx->device));
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	size_t o_seq[2], o_qual[2], o_len[2], o_seed[2], cur = 0;
	for (int m = 0; m < 2; m++) {
		const bt_read_batch* in = m ? in2 : in1;
		o_seq[m] = cur; cur += al((size_t)n * in->stride);
		o_qual[m] = cur; cur += al((size_t)n * in->stride);
		o_len[m] = cur; cur += al(2ull * n);
		o_seed[m] = cur; cur += al(4ull * n);
	}
	const size_t o_hits = cur, o_nh = o_hits + al((size_t)n * out->hit_cap * sizeof(bt_hit)), o_st = o_nh + al(4ull * n),
	             o_mm = o_st + al(n), o_cur = o_mm + al(2ull * out->mm_pool_cap), total = o_cur + 256;
	if (total > c->stage_bytes) {
		if (c->stage) (void)hipFree(c->stage);
		c->stage = nullptr; c->stage_bytes = 0;
		HIPCHK(hipMalloc(&c->stage, total));
		c->stage_bytes = total;
	}
	uint8_t* d = (uint8_t*)c->stage;
	bt_read_batch din[2];
	for (int m = 0; m < 2; m++) {
		const bt_read_batch* in = m ? in2 : in1;
		HIPCHK(hipMemcpyAsync(d + o_seq[m], in->seq, (size_t)n * in->stride, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d + o_qual[m], in->qual, (size_t)n * in->stride, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d + o_len[m], in->len, 2ull * n, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d + o_seed[m], in->seed, 4ull * n, hipMemcpyHostToDevice, c->stream));
		din[m] = *in;
		din[m].seq = d + o_seq[m]; din[m].qual = d + o_qual[m];
		din[m].len = (const uint16_t*)(d + o_len[m]); din[m].seed = (const uint32_t*)(d + o_seed[m]);
	}
	HIPCHK(hipMemsetAsync(d + o_hits, 0, o_cur + 256 - o_hits, c->stream));      /* results, pool and all: what the kernel does not write is zero, not whatever the allocation held */
	bt_hit_batch dout = *out;
	dout.hits = (bt_hit*)(d + o_hits); dout.n_hits = (uint32_t*)(d + o_nh); dout.status = d + o_st;
	dout.mm_pool = out->mm_pool_cap ? (uint16_t*)(d + o_mm) : nullptr;
	if (counts) HIPCHK(hipMemsetAsync(c->d_counts, 0, (CN_N + PS_N) * sizeof(unsigned long long), c->stream));
	rc = run_best_device(c, &din[0], &dout, nullptr, &din[1]);
	if (rc != BT_OK) return rc;
	HIPCHK(hipMemcpyAsync(out->hits, d + o_hits, (size_t)n * out->hit_cap * sizeof(bt_hit), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out->n_hits, d + o_nh, 4ull * n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out->status, d + o_st, n, hipMemcpyDeviceToHost, c->stream));
	rc = bt_ctx_sync(c);
	if (rc != BT_OK) return rc;
	out->mm_pool_used = c->last_mm_used < out->mm_pool_cap ? c->last_mm_used : out->mm_pool_cap;
	/* the entries the kernels handed out, not the pool's capacity */
	if (out->mm_pool_used) {
		HIPCHK(hipMemcpyAsync(out->mm_pool, d + o_mm, 2ull * out->mm_pool_used, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	/* (the on-stream second pass appends to the same pool: a pair searched twice leaves its first lists behind) */
	if ((rc = check_mm_pool(out, n, c->last_mm_used, c->last_dev_retried == 0, c->is_big ? "bt_align_pairs (second pass)" : "bt_align_pairs")) != BT_OK) return rc;
	if (counts) { rc = bt_ctx_counts(c, counts, 0); if (rc != BT_OK) return rc; }
	/* pairs that outgrew their arena: again through the twin context's 16 MB arenas */
	std::vector<uint32_t> redo;
	if (!c->is_big)
		for (uint32_t i = 0; i < n; i++) if (out->status[i] & BT_STF_OVERFLOW) redo.push_back(i);
	c->last_retried = (uint32_t)redo.size();
	if (!redo.empty()) {
		if (!c->big) {
			bt_ctx* b = nullptr;
			rc = bt_ctx_create(c->idx, &c->pol, nullptr, &b);
			if (rc != BT_OK) return rc;
			b->is_big = true;
			c->big = b;
		}
		const uint32_t m = (uint32_t)redo.size();
		std::vector<uint8_t> sseq[2], squal[2]; std::vector<uint16_t> slen[2]; std::vector<uint32_t> sseed[2];
		bt_read_batch sin[2];
		for (int k2 = 0; k2 < 2; k2++) {
			const bt_read_batch* in = k2 ? in2 : in1;
			sseq[k2].resize((size_t)m * in->stride); squal[k2].resize((size_t)m * in->stride); slen[k2].resize(m); sseed[k2].resize(m);
			for (uint32_t k = 0; k < m; k++) {
				const uint32_t i = redo[k];
				memcpy(&sseq[k2][(size_t)k * in->stride], in->seq + (size_t)i * in->stride, in->stride);
				memcpy(&squal[k2][(size_t)k * in->stride], in->qual + (size_t)i * in->stride, in->stride);
				slen[k2][k] = in->len[i]; sseed[k2][k] = in->seed[i];
			}
			sin[k2] = bt_read_batch{ m, in->stride, sseq[k2].data(), squal[k2].data(), slen[k2].data(), sseed[k2].data() };
		}
		std::vector<uint8_t> sst(m); std::vector<uint32_t> snh(m);
		std::vector<bt_hit> shits((size_t)m * out->hit_cap);
		const uint32_t spare = out->mm_pool_cap > out->mm_pool_used ? out->mm_pool_cap - out->mm_pool_used : 0;
		std::vector<uint16_t> spool(spare ? spare : 1);
		bt_hit_batch sout = { out->hit_cap, shits.data(), snh.data(), sst.data(), spool.data(), spare, 0 };
		bt_op_counts c2;
		rc = bt_align_pairs(c->big, &sin[0], &sin[1], &sout, counts ? &c2 : nullptr);
		if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return rc;
		if (counts) {
			counts->lfex += c2.lfex; counts->lf2 += c2.lf2; counts->lf1 += c2.lf1; counts->chase += c2.chase;
			counts->ftab += c2.ftab; counts->offs += c2.offs; counts->rstarts += c2.rstarts; counts->frames += c2.frames;
			counts->same_pair += c2.same_pair;
		}
		for (uint32_t k = 0; k < m; k++) {
			const uint32_t i = redo[k];
			out->n_hits[i] = snh[k]; out->status[i] = sst[k];
			for (uint32_t h = 0; h < out->hit_cap; h++) {
				bt_hit hit = shits[(size_t)k * out->hit_cap + h];
				if (hit.nmm) hit.mm_off += out->mm_pool_used;
				out->hits[(size_t)i * out->hit_cap + h] = hit;
			}
		}
		if (sout.mm_pool_used) memcpy(out->mm_pool + out->mm_pool_used, spool.data(), 2ull * sout.mm_pool_used);
		out->mm_pool_used += sout.mm_pool_used;
	}
	int worst = BT_OK;
	for (uint32_t i = 0; i < n; i++)
		if ((out->status[i] & (BT_STF_OVERFLOW | BT_STF_MMPOOL)) && worst == BT_OK) worst = BT_ERR_OVERFLOW;
	return worst;
}

/* The caller's word that no read of the device-pointer batches to come is longer than max_len (0 = no such promise):
 * the build of the kernel follows from it without a look at the lengths in HBM.  A read that breaks the promise is
 * not searched and comes back flagged BT_ST_OVERFLOW. */
extern "C" int bt_ctx_set_max_read_len(bt_ctx* c, uint32_t max_len)
{
	if (!c) return BT_ERR_ARG;
	c->maxLenHint = max_len;
	return BT_OK;
}

/* Carry-over between the launches of this context (see bt_kernels.h).  On: a bt_align_batch_device call returns
 * its batch's results complete only once the NEXT call's stream work is, or after bt_ctx_sync. */
extern "C" int bt_ctx_set_carry(bt_ctx* c, int launches)
{
	if (!c || launches < 0) return BT_ERR_ARG;
	if (c->carryPending) { const int rc = ctx_flush_carry(c); if (rc != BT_OK) return rc; }
	c->carry = launches != 0;
	c->carryAge = launches > BT_BATCH_RING - 2 ? BT_BATCH_RING - 2 : (uint32_t)launches;
	return BT_OK;
}

/* milliseconds from the start of the first launch since the last bt_ctx_sync to the end of the last one, and how
 * many batches that was (call after bt_ctx_sync) */
extern "C" float bt_ctx_span_ms(bt_ctx* c, uint32_t* n_launches)
{
	float ms = 0.f;
	if (n_launches) *n_launches = c ? c->spanLaunches : 0;
	if (!c || c->spanLaunches == 0) return 0.f;
	if (hipEventElapsedTime(&ms, c->evSpan, c->ev1) != hipSuccess) return -1.f;
	return ms;
}

/* duration of the i-th launch since the last-but-one bt_ctx_sync ... i.e. of the span bt_ctx_span_ms reports (the
 * last 16 are kept); i = -1: the flush launch at the end of the span (0 if there was none).  Call after bt_ctx_sync. */
extern "C" float bt_ctx_launch_ms(bt_ctx* c, int i)
{
	float ms = 0.f;
	if (!c) return 0.f;
	if (i < 0) { if (!c->flushTimed) return 0.f; return hipEventElapsedTime(&ms, c->evFlush[0], c->evFlush[1]) == hipSuccess ? ms : -1.f; }
	if ((uint32_t)i >= c->spanLaunches || (uint32_t)i + 16u < c->spanLaunches) return 0.f;
	return hipEventElapsedTime(&ms, c->evRing[i & 15][0], c->evRing[i & 15][1]) == hipSuccess ? ms : -1.f;
}

extern "C" int bt_ctx_sync(bt_ctx* c)
{
	if (!c) return BT_ERR_ARG;
	if (c->carryPending) { const int rc = ctx_flush_carry(c); if (rc != BT_OK) return rc; }
	HIPCHK(hipStreamSynchronize(c->stream));
	c->spanOpen = false;
	if (c->hostParked) {
		/* diagnostics: reads parked by the launches of the span just closed (all of them: the flush parks none) */
		uint32_t tot = 0;
		for (int i = 0; i < BT_BATCH_RING * BT_BATCH_RING; i++) tot += c->hostParked[i];
		c->last_carried = tot;
		memset(c->hostParked, 0, BT_BATCH_RING * BT_BATCH_RING * 4);
	}
	HIPCHK(hipMemcpy(&c->last_mm_used, c->lastMmCursor ? c->lastMmCursor : c->d_cursor + 1, 4, hipMemcpyDeviceToHost));
	if (!c->is_big) HIPCHK(hipMemcpy(&c->last_dev_retried, c->d_cursor + (c->best ? 2 : 8), 4, hipMemcpyDeviceToHost));
	return BT_OK;
}

extern "C" float bt_ctx_last_kernel_ms(bt_ctx* c)
{
	float ms = 0.f;
	if (!c || !c->timed) return 0.f;
	if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
	return ms;
}

/* profiling build only (-DBT_PROFILE): wavefront cycles per automaton section, PS_N values */
extern "C" const char* bt_ctx_last_kernel_name(bt_ctx* c) { return c ? c->last_kernel : ""; }

extern "C" int bt_ctx_prof_sections(bt_ctx* c, uint64_t* out, int n)
{
	if (!c || !out) return BT_ERR_ARG;
	unsigned long long h[CN_N + PS_N];
	HIPCHK(hipMemcpy(h, c->d_counts, sizeof(h), hipMemcpyDeviceToHost));
	for (int i = 0; i < n && i < PS_N; i++) out[i] = h[CN_N + i];
	return BT_OK;
}

extern "C" void bt_ctx_set_iters_buffer(bt_ctx* c, uint32_t* dev_ptr) { if (c) c->iters_dev = dev_ptr; }

extern "C" uint32_t bt_ctx_last_mm_used(bt_ctx* c) { return c ? c->last_mm_used : 0; }
/* after bt_ctx_counts: the jump table's look-ups among the counted searches, and the LF steps of the reference's algorithm that
 * lay behind them (they are part of bt_op_counts' lf2 / lf1) */
extern "C" void bt_ctx_jump_counts(bt_ctx* c, uint64_t* lookups, uint64_t* steps) { if (lookups) *lookups = c ? c->last_jumps : 0; if (steps) *steps = c ? c->last_jump_steps : 0; }
extern "C" uint64_t bt_index_jump_bytes(const bt_index* idx) { return idx ? idx->jump_bytes : 0; }
/* after bt_ctx_sync: reads the last two launches parked for their successors (carry-over; diagnostics) */
extern "C" uint32_t bt_ctx_last_carried(bt_ctx* c) { return c ? c->last_carried : 0; }
extern "C" uint32_t bt_ctx_last_retried(bt_ctx* c) { return c ? c->last_retried + c->last_dev_retried : 0; }

extern "C" int bt_ctx_counts(bt_ctx* c, bt_op_counts* out, int reset)
{
	if (!c || !out) return BT_ERR_ARG;
	unsigned long long h[CN_N + PS_N];
	HIPCHK(hipMemcpy(h, c->d_counts, sizeof(h), hipMemcpyDeviceToHost));
	out->lfex = h[0]; out->lf2 = h[1]; out->lf1 = h[2]; out->chase = h[3]; out->ftab = h[4];
	out->offs = h[5]; out->rstarts = h[6]; out->frames = h[7];
	out->lane_iters = h[8]; out->same_pair = h[9]; out->rescans = h[10]; out->cand_scans = h[11]; out->wave_rounds = h[12]; out->fetches = h[13];
	/* what locus mode decided by the text is part of the reference's op counts all the same (bt_op_counts) */
	out->loc_lfex = h[CN_TLFEX]; out->loc_lf1 = h[CN_TLF1]; out->loc_chase = h[CN_TCHASE]; out->loc_records = h[CN_LOCREC]; out->loc_windows = h[CN_TXTWIN];
	out->lfex += out->loc_lfex; out->same_pair += out->loc_lfex; out->lf1 += out->loc_lf1; out->chase += out->loc_chase;
	/* ... and so is what lies behind the jump table's look-ups */
	out->lf2 += h[CN_JLF2]; out->lf1 += h[CN_JLF1]; out->same_pair += h[CN_JSAME];
	c->last_jumps = h[CN_JUMPS]; c->last_jump_steps = h[CN_JLF2] + h[CN_JLF1];
	if (reset) HIPCHK(hipMemsetAsync(c->d_counts, 0, sizeof(h), c->stream));      /* ordered before the next launch's tallies */
	return BT_OK;
}

extern "C" int bt_align_batch(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, bt_op_counts* counts)
{
	if (!c || !in || !out) return BT_ERR_ARG;
	const uint32_t n = in->n_reads;
	out->mm_pool_used = 0;
	if (n == 0) return BT_OK;
	if (!in->seq || !in->qual || !in->len || !in->seed || !out->hits || !out->n_hits || !out->status ||
	    out->hit_cap == 0 || (in->stride & 15u) != 0) return BT_ERR_ARG;
	uint32_t maxLen = 0;
	for (uint32_t i = 0; i < n; i++) {
		if (in->len[i] > 1024 || in->len[i] > in->stride) return BT_ERR_ARG;    /* 0 is fine: -3/-5 can trim a read away */
		if (in->len[i] > maxLen) maxLen = in->len[i];
	}
	HIPCHK(hipSetDevice(c->idx->device));
	/* one staging allocation: seq | qual | len | seed | hits | n_hits | status | mm_pool */
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t o_seq = 0, o_qual = o_seq + al((size_t)n * in->stride), o_len = o_qual + al((size_t)n * in->stride),
	             o_seed = o_len + al(2ull * n), o_hits = o_seed + al(4ull * n),
	             o_nh = o_hits + al((size_t)n * out->hit_cap * sizeof(bt_hit)), o_st = o_nh + al(4ull * n),
	             o_mm = o_st + al(n), o_cur = o_mm + al(2ull * out->mm_pool_cap), total = o_cur + 256;
	if (total > c->stage_bytes) {
		if (c->stage) (void)hipFree(c->stage);
		c->stage = nullptr; c->stage_bytes = 0;
		HIPCHK(hipMalloc(&c->stage, total));
		c->stage_bytes = total;
	}
	uint8_t* d = (uint8_t*)c->stage;
	HIPCHK(hipMemcpyAsync(d + o_seq, in->seq, (size_t)n * in->stride, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d + o_qual, in->qual, (size_t)n * in->stride, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d + o_len, in->len, 2ull * n, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(d + o_seed, in->seed, 4ull * n, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemsetAsync(d + o_hits, 0, o_cur + 256 - o_hits, c->stream));      /* results, pool and all: what the kernel does not write is zero, not whatever the allocation held */
	bt_read_batch din = *in;
	din.seq = d + o_seq; din.qual = d + o_qual; din.len = (const uint16_t*)(d + o_len); din.seed = (const uint32_t*)(d + o_seed);
	bt_hit_batch dout = *out;
	dout.hits = (bt_hit*)(d + o_hits); dout.n_hits = (uint32_t*)(d + o_nh); dout.status = d + o_st;
	dout.mm_pool = out->mm_pool_cap ? (uint16_t*)(d + o_mm) : nullptr;
	if (counts) HIPCHK(hipMemsetAsync(c->d_counts, 0, (CN_N + PS_N) * sizeof(unsigned long long), c->stream));
	int rc = run_device(c, &din, &dout, maxLen, nullptr, false, false);
	if (rc != BT_OK) return rc;
	HIPCHK(hipMemcpyAsync(out->hits, d + o_hits, (size_t)n * out->hit_cap * sizeof(bt_hit), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out->n_hits, d + o_nh, 4ull * n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out->status, d + o_st, n, hipMemcpyDeviceToHost, c->stream));
	rc = bt_ctx_sync(c);
	if (rc != BT_OK) return rc;
	out->mm_pool_used = c->last_mm_used < out->mm_pool_cap ? c->last_mm_used : out->mm_pool_cap;
	/* the entries the kernel handed out, not the pool's capacity */
	if (out->mm_pool_used) {
		HIPCHK(hipMemcpyAsync(out->mm_pool, d + o_mm, 2ull * out->mm_pool_used, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	/* (the best-first engine's on-stream second pass appends to the same pool: a read searched twice leaves its first lists behind) */
	if ((rc = check_mm_pool(out, n, c->last_mm_used, c->last_dev_retried == 0, c->is_big ? "bt_align_batch (second pass)" : "bt_align_batch")) != BT_OK) return rc;
	if (counts) { rc = bt_ctx_counts(c, counts, 0); if (rc != BT_OK) return rc; }
	/* reads whose search outgrew the per-read scratch are run again, on the GPU, through a twin
	 * context sized for the worst case (few lanes, huge per-lane arenas) */
	std::vector<uint32_t> redo;
	if (!c->is_big)
		for (uint32_t i = 0; i < n; i++) if (out->status[i] & BT_STF_OVERFLOW) redo.push_back(i);
	c->last_retried = (uint32_t)redo.size();
	if (!redo.empty()) {
		if ((rc = ctx_ensure_big(c, maxLen, nullptr)) != BT_OK) return rc;
		const uint32_t m = (uint32_t)redo.size();
		std::vector<uint8_t> sseq((size_t)m * in->stride), squal((size_t)m * in->stride), sst(m);
		std::vector<uint16_t> slen(m); std::vector<uint32_t> sseed(m), snh(m);
		std::vector<bt_hit> shits((size_t)m * out->hit_cap);
		const uint32_t spare = out->mm_pool_cap > out->mm_pool_used ? out->mm_pool_cap - out->mm_pool_used : 0;
		std::vector<uint16_t> spool(spare ? spare : 1);
		for (uint32_t k = 0; k < m; k++) {
			const uint32_t i = redo[k];
			memcpy(&sseq[(size_t)k * in->stride], in->seq + (size_t)i * in->stride, in->stride);
			memcpy(&squal[(size_t)k * in->stride], in->qual + (size_t)i * in->stride, in->stride);
			slen[k] = in->len[i]; sseed[k] = in->seed[i];
		}
		bt_read_batch sin = { m, in->stride, sseq.data(), squal.data(), slen.data(), sseed.data() };
		bt_hit_batch sout = { out->hit_cap, shits.data(), snh.data(), sst.data(), spool.data(), spare, 0 };
		bt_op_counts c2;
		rc = bt_align_batch(c->big, &sin, &sout, counts ? &c2 : nullptr);
		if (rc != BT_OK && rc != BT_ERR_OVERFLOW && rc != BT_ERR_READ_SHORT) return rc;
		if (counts && c->best) {
			counts->lfex += c2.lfex; counts->lf2 += c2.lf2; counts->lf1 += c2.lf1; counts->chase += c2.chase;
			counts->ftab += c2.ftab; counts->offs += c2.offs; counts->rstarts += c2.rstarts; counts->frames += c2.frames;
			counts->same_pair += c2.same_pair;
		}
		for (uint32_t k = 0; k < m; k++) {
			const uint32_t i = redo[k];
			out->n_hits[i] = snh[k]; out->status[i] = sst[k];
			for (uint32_t h = 0; h < out->hit_cap; h++) {
				bt_hit hit = shits[(size_t)k * out->hit_cap + h];
				if (hit.nmm) hit.mm_off += out->mm_pool_used;
				out->hits[(size_t)i * out->hit_cap + h] = hit;
			}
		}
		if (sout.mm_pool_used) memcpy(out->mm_pool + out->mm_pool_used, spool.data(), 2ull * sout.mm_pool_used);
		out->mm_pool_used += sout.mm_pool_used;
	}
	int worst = BT_OK;
	for (uint32_t i = 0; i < n; i++) {
		if (out->status[i] & BT_STF_TOOSHORT) worst = BT_ERR_READ_SHORT;
		else if ((out->status[i] & (BT_STF_OVERFLOW | BT_STF_MMPOOL)) && worst == BT_OK) worst = BT_ERR_OVERFLOW;
	}
	return worst;
}

static void ctx_free_stream(bt_ctx* c);
/* ---- a stream of host batches ---------------------------------------------------------------------
 * bt_align_batch waits for its batch; a driver that has the next batch ready (a FASTQ reader ahead of
 * the GPU) hands batches over one after the other instead and collects them, in order, as they become
 * complete.  Every batch in flight has a staging area in HBM (recycled); uploads and the results' way
 * back use a copy stream of their own.  With carry-over a batch is complete when none of its reads is
 * parked any more -- the launches report that (BtKernelArgs::parkedOf), the host looks it up here.   */
struct bt_stream_slot {
	void* dev = nullptr; size_t bytes = 0;
	const bt_read_batch* in = nullptr; bt_hit_batch* out = nullptr; void* tag = nullptr;
	size_t o_hits = 0, o_nh = 0, o_st = 0, o_mm = 0;
	uint32_t n = 0; uint32_t mm_used = 0; uint32_t* mmCursor = nullptr;
	hipEvent_t done = nullptr, up = nullptr;
	uint32_t seq = 0, bid = 0; bool carried = false;       /* the carry launch that took it (0: run to completion at once) */
	int state = 0;                       /* 0 free, 1 launched (results not yet on their way back), 2 copy-back enqueued */
};
struct bt_stream {
	std::vector<bt_stream_slot*> slots;                  /* all staging areas */
	std::vector<bt_stream_slot*> inflight;               /* submitted, not yet collected; oldest first */
	hipStream_t copy = nullptr; hipEvent_t searched = nullptr;       /* PCIe traffic runs beside the search, not in its stream */
};

/* results of a batch whose searches have all finished (or are all enqueued, when after_stream): back to the host on
 * the copy stream */
static int stream_copy_back(bt_ctx* c, bt_stream_slot& s, bool after_stream)
{
	uint8_t* d = (uint8_t*)s.dev;
	bt_hit_batch* out = s.out;
	hipStream_t cs = c->hs->copy;
	if (after_stream) {
		HIPCHK(hipEventRecord(c->hs->searched, c->stream));
		HIPCHK(hipStreamWaitEvent(cs, c->hs->searched, 0));
	}
	HIPCHK(hipMemcpyAsync(&s.mm_used, s.mmCursor, 4, hipMemcpyDeviceToHost, cs));
	HIPCHK(hipMemcpyAsync(out->n_hits, d + s.o_nh, 4ull * s.n, hipMemcpyDeviceToHost, cs));
	HIPCHK(hipMemcpyAsync(out->status, d + s.o_st, s.n, hipMemcpyDeviceToHost, cs));
	HIPCHK(hipMemcpyAsync(out->hits, d + s.o_hits, (size_t)s.n * out->hit_cap * sizeof(bt_hit), hipMemcpyDeviceToHost, cs));
	if (out->mm_pool_cap) HIPCHK(hipMemcpyAsync(out->mm_pool, d + s.o_mm, 2ull * out->mm_pool_cap, hipMemcpyDeviceToHost, cs));
	HIPCHK(hipEventRecord(s.done, cs));
	s.state = 2;
	return BT_OK;
}

static void ctx_free_stream(bt_ctx* c)
{
	if (!c->hs) return;
	for (auto* s : c->hs->slots) { if (s->dev) (void)hipFree(s->dev); if (s->done) (void)hipEventDestroy(s->done); if (s->up) (void)hipEventDestroy(s->up); delete s; }
	if (c->hs->copy) (void)hipStreamDestroy(c->hs->copy);
	if (c->hs->searched) (void)hipEventDestroy(c->hs->searched);
	delete c->hs;
	c->hs = nullptr;
}

extern "C" int bt_align_stream_submit(bt_ctx* c, const bt_read_batch* in, bt_hit_batch* out, void* tag)
{
	if (!c || !in || !out || in->n_reads == 0) return BT_ERR_ARG;
	const uint32_t n = in->n_reads;
	if (!in->seq || !in->qual || !in->len || !in->seed || !out->hits || !out->n_hits || !out->status ||
	    out->hit_cap == 0 || (in->stride & 15u) != 0) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	if (!c->hs) {
		c->hs = new bt_stream();
		HIPCHK(hipStreamCreateWithFlags(&c->hs->copy, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&c->hs->searched, hipEventDisableTiming));
	}
	bt_stream& S = *c->hs;
	/* a ring slot is reused after BT_BATCH_RING launches: nothing that old may still be in flight */
	if (S.inflight.size() >= BT_BATCH_RING - 2u) return BT_ERR_ARG;  /* collect (or flush) first */
	uint32_t maxLen = 0;
	for (uint32_t i = 0; i < n; i++) {
		if (in->len[i] > 1024 || in->len[i] > in->stride) return BT_ERR_ARG;
		if (in->len[i] > maxLen) maxLen = in->len[i];
	}
	bt_stream_slot* sp = nullptr;
	for (auto* x : S.slots) if (x->state == 0) { sp = x; break; }
	if (!sp) { sp = new bt_stream_slot(); S.slots.push_back(sp); }
	bt_stream_slot& s = *sp;
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t o_seq = 0, o_qual = o_seq + al((size_t)n * in->stride), o_len = o_qual + al((size_t)n * in->stride),
	             o_seed = o_len + al(2ull * n), o_hits = o_seed + al(4ull * n),
	             o_nh = o_hits + al((size_t)n * out->hit_cap * sizeof(bt_hit)), o_st = o_nh + al(4ull * n),
	             o_mm = o_st + al(n), o_cur = o_mm + al(2ull * out->mm_pool_cap), total = o_cur + 256;
	if (total > s.bytes) {
		if (s.dev) (void)hipFree(s.dev);
		s.dev = nullptr; s.bytes = 0;
		HIPCHK(hipMalloc(&s.dev, total));
		s.bytes = total;
	}
	if (!s.done) { HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s.up, hipEventDisableTiming)); }
	uint8_t* d = (uint8_t*)s.dev;
	hipStream_t cs = S.copy;
	HIPCHK(hipMemcpyAsync(d + o_seq, in->seq, (size_t)n * in->stride, hipMemcpyHostToDevice, cs));
	HIPCHK(hipMemcpyAsync(d + o_qual, in->qual, (size_t)n * in->stride, hipMemcpyHostToDevice, cs));
	HIPCHK(hipMemcpyAsync(d + o_len, in->len, 2ull * n, hipMemcpyHostToDevice, cs));
	HIPCHK(hipMemcpyAsync(d + o_seed, in->seed, 4ull * n, hipMemcpyHostToDevice, cs));
	HIPCHK(hipMemsetAsync(d + o_hits, 0, o_mm - o_hits, cs));
	/* diagnostics: BT_STREAM_POISON=1 fills the mismatch pool's staging region with 0xff, so that an entry the
	 * host reads without the kernel having written it is recognisable (position 0x3ff, base 3) instead of looking like another
	 * read's entry left over from whoever had the memory before */
	if (ctx_env(c, "BT_STREAM_POISON", 0) && out->mm_pool_cap) HIPCHK(hipMemsetAsync(d + o_mm, 0xff, 2ull * out->mm_pool_cap, cs));
	HIPCHK(hipEventRecord(s.up, cs));
	HIPCHK(hipStreamWaitEvent(c->stream, s.up, 0));
	bt_read_batch din = *in;
	din.seq = d + o_seq; din.qual = d + o_qual; din.len = (const uint16_t*)(d + o_len); din.seed = (const uint32_t*)(d + o_seed);
	bt_hit_batch dout = *out;
	dout.hits = (bt_hit*)(d + o_hits); dout.n_hits = (uint32_t*)(d + o_nh); dout.status = d + o_st;
	dout.mm_pool = out->mm_pool_cap ? (uint16_t*)(d + o_mm) : nullptr;
	const uint32_t seq0 = c->launchSeq;
	const bool wasPending = c->carryPending;
	/* reads that outgrow their scratch stay flagged (BT_ST_OVERFLOW) for the caller: no second pass on the stream here */
	/* the mismatch-pool cursor lives in the staging area: the context's own (d_cursor[1]) is reset by the next launch, which
	 * may be enqueued before this batch's results have been copied */
	/* BT_STREAM_OLD_CURSOR=1 (diagnostics): the cursors rounds 2-3 used -- the ring's for a carried batch, the
	 * context's otherwise */
	const int rc = run_device(c, &din, &dout, maxLen, nullptr, false, true, false, ctx_env(c, "BT_STREAM_OLD_CURSOR", 0) ? nullptr : (uint32_t*)(d + o_cur));
	if (rc != BT_OK) return rc;
	s.in = in; s.out = out; s.tag = tag; s.n = n; s.o_hits = o_hits; s.o_nh = o_nh; s.o_st = o_st; s.o_mm = o_mm;
	s.mmCursor = c->lastMmCursor; s.state = 1;
	const bool carriedNow = c->carryPending;                       /* run_device parked this batch's last reads */
	const bool flushed = wasPending && (c->launchSeq - seq0) == (carriedNow ? 2u : 1u);   /* ... after finishing what was parked before */
	if (flushed) for (auto* x : S.inflight) if (x->state == 1) { x->carried = false; const int r2 = stream_copy_back(c, *x, true); if (r2 != BT_OK) return r2; }
	S.inflight.push_back(sp);
	if (!carriedNow) { s.carried = false; return stream_copy_back(c, s, true); }
	s.carried = true; s.seq = c->launchSeq - 1u; s.bid = s.seq & (BT_BATCH_RING - 1u);
	return BT_OK;
}

/* The oldest submitted batch, if it is complete: its bt_hit_batch is filled (mm_pool_used included) and *tag is what
 * came with it.  *tag = NULL with BT_OK: nothing in flight, or (flush == 0) the oldest batch still has reads being
 * searched -- submit more, or ask again later.  flush != 0: nothing follows, finish whatever is parked now. */
extern "C" int bt_align_stream_collect(bt_ctx* c, void** tag, int flush)
{
	if (!c || !tag) return BT_ERR_ARG;
	*tag = nullptr;
	if (!c->hs || c->hs->inflight.empty()) return BT_OK;
	bt_stream& S = *c->hs;
	HIPCHK(hipSetDevice(c->idx->device));
	bt_stream_slot& s = *S.inflight.front();
	if (s.state == 1) {
		bool complete = false;
		if (s.carried && !flush) {
			/* the latest finished launch since the batch's own says how many of its reads are still parked */
			for (uint32_t q = c->launchSeq - 1u; q >= s.seq; q--) {          /* seq >= 1 */
				const uint32_t k = q & (BT_BATCH_RING - 1u);
				if (hipEventQuery(c->evLaunch[k]) == hipSuccess) { complete = c->hostParked[(size_t)k * BT_BATCH_RING + s.bid] == 0; break; }
			}
			if (!complete) return BT_OK;
			const int r2 = stream_copy_back(c, s, ctx_env(c, "BT_STREAM_ORDERED", 0) != 0);
			if (r2 != BT_OK) return r2;
			/* The copies are enqueued; they run when the launch that is searching lets them (on this device a copy does not
			 * start beside the persistent kernel: it waits for the launch's end, whatever stream, priority or memory it uses --
			 * round 6, GPU calls 14-16).  Waiting for them here kept the caller from submitting for that long, 0.75 s per
			 * collected batch, and the stream ran dry every few batches of a long run: ask again later instead (the caller
			 * polls; the batch is in state "copying" and the next call looks at its event only). */
			if (!ctx_env(c, "BT_STREAM_COLLECT_WAITS", 0) && hipEventQuery(s.done) != hipSuccess) { (void)hipGetLastError(); return BT_OK; }
		} else {
			if (!flush) return BT_ERR_ARG;                          /* cannot happen: uncarried batches are copied back at submit */
			const int rc = ctx_flush_carry(c);
			if (rc != BT_OK) return rc;
			for (auto* x : S.inflight) if (x->state == 1) { const int r2 = stream_copy_back(c, *x, true); if (r2 != BT_OK) return r2; }
		}
	} else if (!flush && hipEventQuery(s.done) != hipSuccess) return BT_OK;
	HIPCHK(hipEventSynchronize(s.done));
	s.out->mm_pool_used = s.mm_used < s.out->mm_pool_cap ? s.mm_used : s.out->mm_pool_cap;
	/* the batch's hit records against its own pool cursor (one pass, one cursor, whatever launches its reads rode along with) */
	{ const int rcm = check_mm_pool(s.out, s.n, s.mm_used, true, "bt_align_stream_collect"); if (rcm != BT_OK) return rcm; }
	if (ctx_env(c, "BT_STREAM_RECHECK", 0)) {
		/* diagnostics: with the device idle, what the staging area holds now against what the copy stream
		 * delivered -- a difference means the copy ran before the batch's last writes were there */
		HIPCHK(hipDeviceSynchronize());
		const uint8_t* d = (const uint8_t*)s.dev;
		std::vector<uint8_t> t;
		auto cmp = [&](const char* what, const void* host, size_t off, size_t bytes) -> int {
			t.resize(bytes);
			if (bytes == 0) return 0;
			if (hipMemcpy(t.data(), d + off, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
			size_t nd = 0, first = 0;
			for (size_t i = 0; i < bytes; i++) if (t[i] != ((const uint8_t*)host)[i]) { if (!nd) first = i; nd++; }
			if (nd) fprintf(stderr, "[stream-recheck] tag %p carried=%d seq=%u bid=%u: %s differs in %zu of %zu bytes (first at %zu): delivered early\n",
			                s.tag, (int)s.carried, s.seq, s.bid, what, nd, bytes, first);
			return nd ? 1 : 0;
		};
		int bad = 0;
		bad |= cmp("n_hits", s.out->n_hits, s.o_nh, 4ull * s.n);
		bad |= cmp("status", s.out->status, s.o_st, s.n);
		bad |= cmp("hits", s.out->hits, s.o_hits, (size_t)s.n * s.out->hit_cap * sizeof(bt_hit));
		bad |= cmp("mm_pool", s.out->mm_pool, s.o_mm, 2ull * s.out->mm_pool_used);
		uint32_t cur = 0;
		if (hipMemcpy(&cur, s.mmCursor, 4, hipMemcpyDeviceToHost) == hipSuccess && s.carried && cur != s.mm_used)
			fprintf(stderr, "[stream-recheck] tag %p: mm cursor %u now, %u delivered\n", s.tag, cur, s.mm_used);
		if (bad > 0 && ctx_env(c, "BT_STREAM_RECHECK", 0) > 1) return BT_ERR_DEVICE;
	}
	*tag = s.tag;
	s.state = 0;
	S.inflight.erase(S.inflight.begin());
	return BT_OK;
}

extern "C" int bt_align_stream_tick(bt_ctx* c, uint32_t min_rounds)
{
	if (!c) return BT_ERR_ARG;
	if (!c->carryPending || !c->pool) return BT_OK;              /* nothing is parked */
	HIPCHK(hipSetDevice(c->idx->device));
	/* the launch of ctx_flush_carry, except that it parks again: same grid, no fresh reads, a ring slot of its own whose
	 * batch is empty (a slot is reused BT_BATCH_RING launches later; what used it has been complete for two at least by then:
	 * nothing is carried through more than BT_BATCH_RING - 2 launches) */
	BtKernelArgs A;
	memset(&A, 0, sizeof(A));
	BtWarm warm;
	fill_index_args(c, &A, &warm);
	const uint32_t bid = c->launchSeq & (BT_BATCH_RING - 1u);
	BatchView none;
	memset(&none, 0, sizeof(none));
	c->ring[bid] = none; c->ringRetry[bid] = false; c->ringMaxLen[bid] = 0;
	BtCold cold;
	fill_cold(c, &cold, none.B, bid);
	BT_H2D(c->d_cold, &cold, sizeof(cold));
	BT_H2D(c->d_warm, &warm, sizeof(warm));
	const uint32_t init[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	BT_H2D(c->d_cursor, init, sizeof(init));
	HIPCHK(hipMemsetAsync(c->d_carry, 0, BT_BATCH_RING * 4, c->stream));                 /* this launch's parked counts */
	A.H.seq = nullptr; A.H.qual = nullptr; A.H.stride = 0; A.H.n_reads = 0;
	A.cold = c->d_cold; A.warm = c->d_warm;
	A.frames = c->frames; A.pairs = c->pairs; A.meta = c->meta; A.pals = c->pals;
	A.nLanes = c->nLanes; A.nSlots = c->nSlots; A.frCap = c->frCap; A.entCap = c->entCap; A.palCap = c->palCap;
	A.counts = c->d_counts;
	A.nextRead = c->d_cursor;
	A.pool = c->pool; A.launchSeq = c->launchSeq; A.adopt = 1; A.park = 1; A.maxAge = c->carryAge; A.parkedOf = c->d_carry;
	A.parkMinRounds = min_rounds ? min_rounds : ctx_env(c, "BT_TICK_MIN_ROUNDS", 150000);
	if (bt_launch_search(&A, c->carryBlocks, c->occ, c->carryRl | (ctx_env(c, "BT_FORCE_EXT", 0) ? BT_RL_FORCE_EXT : 0), c->stream) != 0) return BT_ERR_DEVICE;
	const uint32_t k = c->launchSeq & (BT_BATCH_RING - 1u);
	HIPCHK(hipMemcpyAsync(c->hostParked + (size_t)k * BT_BATCH_RING, c->d_carry, BT_BATCH_RING * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipEventRecord(c->evLaunch[k], c->stream));
	c->launchSeq++;
	return BT_OK;
}

/* How many batches of this shape the stream may hold in flight as far as the device's memory goes: a batch in flight has a
 * staging area in HBM (reads in, results out: the layout of bt_align_stream_submit), and a caller that lets two dozen of them
 * ride (bowtie-amd on a large host) must not find out at the twentieth that the device is full -- a large -k makes the area
 * several times the reads' size.  Half of what is free now, plus the areas this context already has and is not using. */
extern "C" int bt_align_stream_room(bt_ctx* c, const bt_read_batch* in, const bt_hit_batch* out, uint32_t* batches)
{
	if (!c || !in || !out || !batches || in->n_reads == 0 || out->hit_cap == 0) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t n = in->n_reads;
	const size_t area = 2u * al(n * in->stride) + al(2ull * n) + al(4ull * n) + al(n * out->hit_cap * sizeof(bt_hit)) + al(4ull * n) + al(n) +
	                    al(2ull * out->mm_pool_cap) + 256u;
	size_t freeB = 0, totB = 0;
	if (hipMemGetInfo(&freeB, &totB) != hipSuccess) { (void)hipGetLastError(); return BT_ERR_DEVICE; }
	{ const size_t v = (size_t)ctx_env(c, "BT_FAKE_FREE_MB", 0) << 20; if (v && v < freeB) freeB = v; }   /* (tests) */
	uint64_t fit = (uint64_t)(freeB / 2u) / area;
	if (c->hs) for (auto* x : c->hs->slots) if (x->state == 0 && x->bytes >= area) fit++;
	*batches = fit > 0xffffffffull ? 0xffffffffu : (uint32_t)fit;
	return BT_OK;
}

extern "C" void* bt_host_alloc(size_t bytes)
{
	void* p = nullptr;
	if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess /* any thread, any device */) { (void)hipGetLastError(); return nullptr; }
	return p;
}
extern "C" void bt_host_free(void* p) { if (p) (void)hipHostFree(p); }

/* ---- probes ---------------------------------------------------------------------------------- */
/* rows as 64-bit numbers in either build (the wide build's own probe; the 32-bit calls below answer BT_ERR_UNSUPPORTED there) */
extern "C" int bt_probe_rank64(bt_ctx* c, int mirror, const uint64_t* rows, uint32_t n, uint64_t* lf, uint8_t* L)
{
	if (!c || !rows || !lf || !L || (mirror && !c->idx->has_mirror)) return BT_ERR_ARG;
	if (n == 0) return BT_OK;
	HIPCHK(hipSetDevice(c->idx->device));
	std::vector<bt_row> hr(n), hl(4ull * n);
	for (uint32_t i = 0; i < n; i++) hr[i] = (bt_row)rows[i];
	bt_row *d_rows = nullptr, *d_lf = nullptr; uint8_t* d_L = nullptr;
	HIPCHK(hipMalloc((void**)&d_rows, sizeof(bt_row) * (size_t)n)); HIPCHK(hipMalloc((void**)&d_lf, 4u * sizeof(bt_row) * (size_t)n)); HIPCHK(hipMalloc((void**)&d_L, n));
	HIPCHK(hipMemcpy(d_rows, hr.data(), sizeof(bt_row) * (size_t)n, hipMemcpyHostToDevice));
	int rc = bt_launch_probe_rank(&c->idx->dev[mirror ? 1 : 0], d_rows, n, d_lf, d_L, 0, c->stream);
	if (rc == 0) rc = (int)hipStreamSynchronize(c->stream);
	if (rc == 0) { (void)hipMemcpy(hl.data(), d_lf, 4u * sizeof(bt_row) * (size_t)n, hipMemcpyDeviceToHost); (void)hipMemcpy(L, d_L, n, hipMemcpyDeviceToHost); }
	for (size_t i = 0; i < 4ull * n; i++) lf[i] = hl[i];
	(void)hipFree(d_rows); (void)hipFree(d_lf); (void)hipFree(d_L);
	return rc == 0 ? BT_OK : BT_ERR_DEVICE;
}

#if BT_WIDE
extern "C" int bt_probe_rank(bt_ctx*, int, const uint32_t*, uint32_t, uint32_t*, uint8_t*) { return BT_ERR_UNSUPPORTED; }
extern "C" int bt_probe_chase(bt_ctx*, int, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*) { return BT_ERR_UNSUPPORTED; }
extern "C" int bt_bench_gather(bt_ctx*, int, uint32_t, uint32_t, int, float*, double*) { return BT_ERR_UNSUPPORTED; }
#else
extern "C" int bt_probe_rank(bt_ctx* c, int mirror, const uint32_t* rows, uint32_t n, uint32_t* lf, uint8_t* L)
{
	const uint32_t sides = (uint32_t)mirror & 2u;          /* bit 1: rank from the side layout (see the header) */
	mirror &= 1;
	if (!c || !rows || !lf || !L || (mirror && !c->idx->has_mirror)) return BT_ERR_ARG;
	if (n == 0) return BT_OK;
	HIPCHK(hipSetDevice(c->idx->device));
	uint32_t *d_rows = nullptr, *d_lf = nullptr; uint8_t* d_L = nullptr;
	HIPCHK(hipMalloc((void**)&d_rows, 4ull * n)); HIPCHK(hipMalloc((void**)&d_lf, 16ull * n)); HIPCHK(hipMalloc((void**)&d_L, n));
	HIPCHK(hipMemcpy(d_rows, rows, 4ull * n, hipMemcpyHostToDevice));
	int rc = bt_launch_probe_rank(&c->idx->dev[mirror ? 1 : 0], d_rows, n, d_lf, d_L, sides, c->stream);
	if (rc == 0) rc = (int)hipStreamSynchronize(c->stream);
	if (rc == 0) { (void)hipMemcpy(lf, d_lf, 16ull * n, hipMemcpyDeviceToHost); (void)hipMemcpy(L, d_L, n, hipMemcpyDeviceToHost); }
	(void)hipFree(d_rows); (void)hipFree(d_lf); (void)hipFree(d_L);
	return rc == 0 ? BT_OK : BT_ERR_DEVICE;
}

extern "C" int bt_probe_chase(bt_ctx* c, int mirror, const uint32_t* rows, uint32_t n, uint32_t qlen,
                              uint32_t* joined_off, uint32_t* tidx, uint32_t* toff)
{
	if (!c || !rows || !joined_off || !tidx || !toff || (mirror && !c->idx->has_mirror)) return BT_ERR_ARG;
	if (n == 0) return BT_OK;
	HIPCHK(hipSetDevice(c->idx->device));
	uint32_t* d = nullptr;
	HIPCHK(hipMalloc((void**)&d, 16ull * n));
	HIPCHK(hipMemcpy(d, rows, 4ull * n, hipMemcpyHostToDevice));
	int rc = bt_launch_probe_chase(&c->idx->dev[mirror ? 1 : 0], d, n, qlen, d + n, d + 2ull * n, d + 3ull * n, c->stream);
	if (rc == 0) rc = (int)hipStreamSynchronize(c->stream);
	if (rc == 0) {
		(void)hipMemcpy(joined_off, d + n, 4ull * n, hipMemcpyDeviceToHost);
		(void)hipMemcpy(tidx, d + 2ull * n, 4ull * n, hipMemcpyDeviceToHost);
		(void)hipMemcpy(toff, d + 3ull * n, 4ull * n, hipMemcpyDeviceToHost);
	}
	(void)hipFree(d);
	return rc == 0 ? BT_OK : BT_ERR_DEVICE;
}

/* Random-128-byte-gather ceiling: n_blocks x 256 threads, `iters` rank queries each, timed with HIP
 * events; *gbs = queries x 128 B / time. */
extern "C" int bt_bench_gather(bt_ctx* c, int mirror, uint32_t n_blocks, uint32_t iters, int dependent,
                               float* ms_out, double* gbs_out)
{
	const bool sides = (mirror & 2) != 0;                  /* bit 1: gather 128-byte side pairs instead of 32-byte rank blocks */
	mirror &= 1;
	if (!c || !ms_out || !gbs_out || n_blocks == 0 || iters == 0 || (mirror && !c->idx->has_mirror)) return BT_ERR_ARG;
	HIPCHK(hipSetDevice(c->idx->device));
	uint32_t* sink = c->d_cursor + 7;
	for (int rep = 0; rep < 2; rep++) {            /* first pass warms the TLBs */
		HIPCHK(hipEventRecord(c->ev0, c->stream));
		if (bt_launch_gather_bench(&c->idx->dev[mirror ? 1 : 0], n_blocks, iters, (dependent ? 1u : 0u) | (sides ? 2u : 0u), sink, c->stream) != 0) return BT_ERR_DEVICE;
		HIPCHK(hipEventRecord(c->ev1, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	float ms = 0.f;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	*ms_out = ms;
	*gbs_out = (double)n_blocks * 256.0 * iters * (sides ? 128.0 : 32.0) / (ms * 1e-3) / 1e9;
	return BT_OK;
}
#endif /* BT_WIDE */

extern "C" int bt_index_restore_text(const char* ebwt_base, uint8_t* out, uint64_t cap)
{
	if (!ebwt_base || !out) return BT_ERR_ARG;
	BtIndexHost h;
	int rc;
	try { rc = bt_host_index_load(ebwt_base, true, -1, &h); } catch (const std::exception&) { rc = BT_ERR_FORMAT; }
	if (rc != BT_OK) return rc;
	if (cap < h.len) return BT_ERR_ARG;
	bt_host_restore_text(h, out);
	return BT_OK;
}

/* Host-side: what an index base holds once loaded -- the variant found on disk, the text length, and a digest
 * (FNV-1a 64) of each array of the in-memory image.  Two bases that load to the same image (a .ebwtl, .bt2 or
 * other-endian copy of the same index) give the same digests. */
extern "C" int bt_index_digest(const char* ebwt_base, int mirror, uint64_t out[8])
{
	if (!ebwt_base || !out) return BT_ERR_ARG;
	BtIndexHost h;
	const int variant = bt_host_index_variant(ebwt_base);
	if (variant < 0) return BT_ERR_IO;
	int rc;
	try { rc = bt_host_index_load(std::string(ebwt_base) + (mirror ? ".rev" : ""), !mirror, -1, &h, variant); }
	catch (const std::exception&) { rc = BT_ERR_FORMAT; }
	if (rc != BT_OK) return rc;
	auto fnv = [](const void* p, size_t n, uint64_t hsh = 1469598103934665603ull) {
		const uint8_t* b = (const uint8_t*)p;
		for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; }
		return hsh;
	};
	out[0] = (uint64_t)variant | (h.swapped ? 16u : 0u);
	out[1] = h.len;
#if BT_WIDE
	out[2] = fnv(h.blk.data(), h.blk.size());          /* (the wide build's image: rank blocks instead of sides, 8-byte entries) */
#else
	out[2] = fnv(h.ebwt.data(), h.ebwt.size());
#endif
	out[3] = fnv(h.ftab.data(), sizeof(bt_row) * h.ftab.size());
	out[4] = fnv(h.eftab.data(), sizeof(bt_row) * h.eftab.size());
	out[5] = fnv(h.offs.data(), sizeof(bt_row) * h.offs.size());
	out[6] = fnv(h.rstarts.data(), sizeof(bt_row) * h.rstarts.size(), fnv(h.plen.data(), sizeof(bt_row) * h.plen.size()));
	bt_row tail[8] = {h.zOff, h.fchr[0], h.fchr[1], h.fchr[2], h.fchr[3], h.fchr[4], (bt_row)h.offRate, (bt_row)h.ftabChars};
	out[7] = fnv(tail, sizeof(tail));
	return BT_OK;
}

/* Host-side, a few bytes of <base>.1.<ext> read: 1 = the index has 2^32-1 rows or more (libbowtie_amd_l.so's job),
 * 0 = it fits 32-bit rows, < 0 = BT_ERR_IO / BT_ERR_FORMAT. */
extern "C" int bt_index_needs_rows64(const char* ebwt_base)
{
	if (!ebwt_base) return BT_ERR_ARG;
	uint64_t len = 0;
	const int rc = bt_host_index_header_len(ebwt_base, &len);
	if (rc != BT_OK) return rc;
	return len >= 0xffffffffull ? 1 : 0;
}

extern "C" const char* bt_strerror(int code)
{
	switch (code) {
	case BT_OK: return "ok";
	case BT_ERR_IO: return "index file missing or truncated";
	case BT_ERR_FORMAT: return "not a bowtie index this build can hold (.ebwt/.ebwtl/.bt2/.bt2l with fewer than 2^32-1 rows)";
	case BT_ERR_ARG: return "bad argument";
	case BT_ERR_DEVICE: return "HIP device error";
	case BT_ERR_READ_SHORT: return "read shorter than the alignment mode allows";
	case BT_ERR_OVERFLOW: return "per-read scratch capacity exceeded";
	case BT_ERR_READS: return "malformed read input";
	case BT_ERR_ROWS64: return "the index has 2^32-1 rows or more: use the build with 64-bit rows (libbowtie_amd_l.so, bowtie-amd-l)";
	case BT_ERR_UNSUPPORTED: return "not in this build (the build with 64-bit rows has bt_probe_rank64 instead of the 32-bit probes, and no gather benchmark)";
	default: return "unknown error";
	}
}
extern "C" const char* bt_version(void) { return BT_WIDE ? "bowtie_amd 0.2.0 (gfx950, 64-bit rows)" : "bowtie_amd 0.2.0 (gfx950)"; }
