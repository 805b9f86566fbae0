/*
 * bt_emu.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Host build of the per-read automaton in bowtie_amd/csrc/bt_core.h, driven the way the HIP
 * kernel drives it (lock-step lanes, LF requests answered between steps), so the automaton's
 * logic can be checked against the oracle in a container that has no GPU.  It is not part of
 * libbowtie_amd.so and nothing in the product loads it; GPU parity is tested separately
 * (tests/test_gpu_*.py) through the C ABI.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../bowtie_amd/csrc/bt_host.h"
/* EMU_MSAN (tests/emu/emu_msan.cpp, clang -fsanitize=memory): everything the kernel leaves undefined -- LDS, the
 * scratch arenas, the parts of a round's result nothing wrote -- is poisoned, so a read of it is reported */
#ifdef EMU_MSAN
#include <sanitizer/msan_interface.h>
#define EMU_POISON(p, n) __msan_poison((p), (n))
#else
#define EMU_POISON(p, n) ((void)0)
#endif

#ifdef BT_L2_DEBUG
unsigned long long g_l2_hit = 0, g_l2_miss = 0, g_l2_none = 0;
extern "C" void emu_l2_stats(unsigned long long* o) { o[0] = g_l2_hit; o[1] = g_l2_miss; o[2] = g_l2_none; }
#endif
struct EmuIndex { BtIndexHost h[2]; BtIndexDev d[2]; std::vector<uint8_t> blk[2]; std::vector<BtU4> loc[2]; std::vector<uint32_t> rtxt[2]; std::vector<uint16_t> walk[2]; std::vector<uint32_t> jump[2]; std::vector<uint16_t> jumpMeta[2]; bool mirror; std::string base; BtRefHost ref; BtRefDev refd; bool haveRef = false; };

static void bind(EmuIndex* e, int m)
{
	bt_host_index_describe(e->h[m], &e->d[m]);
	e->d[m].ebwt = e->h[m].ebwt.data(); e->d[m].ftab = e->h[m].ftab.data(); e->d[m].eftab = e->h[m].eftab.data();
	e->d[m].offs = e->h[m].offs.data(); e->d[m].rstarts = e->h[m].rstarts.data(); e->d[m].plen = e->h[m].plen.data();
#if BT_WIDE
	/* the wide build (64-bit rows): the loader made the rank blocks and their segment table; no locus image */
	e->d[m].ebwt = nullptr;
	e->d[m].blk = e->h[m].blk.data(); e->d[m].segBase = e->h[m].segBase.data();
	e->d[m].loc = nullptr; e->d[m].rtxt = nullptr; e->d[m].walk = nullptr;
	bt_host_index_bias(e->h[m], &e->d[m]);
	return;
#else
	/* the rank blocks the search queries (bt_rank.h), derived from the sides as the GPU loader derives them */
	e->blk[m].assign((size_t)bt_blk_count(e->d[m].len) * BT_BLK_BYTES, 0);
	bt_blk_build_host(e->d[m], e->blk[m].data());
	e->d[m].blk = e->blk[m].data();
	/* the locus image (bt_rank.h), as the GPU loader derives it; EMU_LOCUS=0: an index without one (row space only) */
	e->d[m].loc = nullptr; e->d[m].rtxt = nullptr; e->d[m].walk = nullptr;
	if (!getenv("EMU_LOCUS") || atoi(getenv("EMU_LOCUS")) != 0) {
		e->loc[m].assign((size_t)e->d[m].len + 1u, BtU4{0, 0, 0, 0});
		e->rtxt[m].assign((size_t)bt_rtxt_words(e->d[m].len), 0u);
		e->walk[m].assign((size_t)e->d[m].len + 1u, 0);
		bt_loc_build_host(e->d[m], e->loc[m].data(), e->rtxt[m].data(), e->walk[m].data());
		e->d[m].loc = e->loc[m].data(); e->d[m].rtxt = e->rtxt[m].data() + BT_RTXT_PAD_WORDS; e->d[m].walk = e->walk[m].data();
	}
	/* the jump table (bt_rank.h), as the GPU loader derives it -- for fewer characters, so that the host's copy stays small:
	 * ftabChars + 2 of them (at most 12) for a genome of 32 K bases and more, EMU_JUMP_CHARS=<n> for any index (0: none) */
	e->d[m].jump = nullptr; e->d[m].jumpMeta = nullptr; e->d[m].jumpChars = 0;
	{
		uint32_t K = e->d[m].len >= (1u << 15) ? (e->d[m].ftabChars + 2u > 12u ? 12u : e->d[m].ftabChars + 2u) : 0u;
		if (const char* ev = getenv("EMU_JUMP_CHARS")) K = (uint32_t)atoi(ev);
		if (K > e->d[m].ftabChars && K <= 13u && K - e->d[m].ftabChars <= 7u) {
			const size_t n = (size_t)1 << (2u * K);
			e->jump[m].assign(2u * n, 0u); e->jumpMeta[m].assign(n + 8u, 0);
			for (size_t x = 0; x < n; x++) {
				uint32_t top, bot, meta;
				bt_jump_entry(e->d[m], (uint32_t)x, K, &top, &bot, &meta);
				e->jump[m][2u * x] = top; e->jump[m][2u * x + 1u] = bot; e->jumpMeta[m][x] = (uint16_t)meta;
			}
			e->d[m].jump = e->jump[m].data(); e->d[m].jumpMeta = e->jumpMeta[m].data(); e->d[m].jumpChars = K;
		}
	}
#endif
}

extern "C" void* emu_index_load(const char* base, int need_mirror, int offrate)
{
	EmuIndex* e = new EmuIndex();
	e->mirror = need_mirror != 0;
	e->base = base;
	if (bt_host_index_load(base, true, offrate, &e->h[0]) != BT_OK) { delete e; return nullptr; }
	e->h[0].ebwt.resize(e->h[0].ebwt.size() + 128); e->h[0].ftab.resize(e->h[0].ftab.size() + 4); e->h[0].offs.resize(e->h[0].offs.size() + 4);
	bind(e, 0);
	if (need_mirror) {
		if (bt_host_index_load(std::string(base) + ".rev", false, offrate, &e->h[1]) != BT_OK) { delete e; return nullptr; }
		e->h[1].ebwt.resize(e->h[1].ebwt.size() + 128); e->h[1].ftab.resize(e->h[1].ftab.size() + 4); e->h[1].offs.resize(e->h[1].offs.size() + 4);
		bind(e, 1);
	} else e->d[1] = e->d[0];
	return e;
}
extern "C" void emu_index_free(void* p) { delete (EmuIndex*)p; }
/* the host-built locus image of one index (tests compare the GPU loader's with it) */
extern "C" int emu_locus_arrays(void* p, int mirror, const void** loc, const void** rtxt, const void** walk, uint32_t* len)
{
	EmuIndex* e = (EmuIndex*)p;
	const int m = mirror ? 1 : 0;
	if (!e->d[m].loc) return 1;
	*loc = e->d[m].loc; *rtxt = e->d[m].rtxt; *walk = e->d[m].walk; *len = (uint32_t)e->d[m].len;
	return 0;
}

#if !BT_WIDE
/* the image's pass 1 as the GPU loader runs it (bt_rank.h: bt_loc_chain, a chain per sampled row) against the host build's
 * one walk over the text: 0 = the SA column, the reversed text and the walk lengths are the same, else which differs */
extern "C" int emu_locus_chains_check(void* p, int mirror)
{
	EmuIndex* e = (EmuIndex*)p;
	const int m = mirror ? 1 : 0;
	const BtIndexDev& ix = e->d[m];
	if (!ix.loc) return -1;
	std::vector<BtU4> loc((size_t)ix.len + 1u, BtU4{0xdeadbeefu, 0, 0, 0});
	std::vector<uint32_t> rtxt((size_t)bt_rtxt_words(ix.len), 0u);
	std::vector<uint16_t> walk((size_t)ix.len + 1u, 0xbeef);
	const uint64_t n = bt_loc_chain_count(ix);
	for (uint64_t i = 0; i < n; i++) bt_loc_chain(ix, i, loc.data(), rtxt.data() + BT_RTXT_PAD_WORDS, walk.data());
	for (size_t r = 0; r <= ix.len; r++) if (loc[r].x != e->loc[m][r].x) return 1;
	if (memcmp(rtxt.data(), e->rtxt[m].data(), rtxt.size() * 4u) != 0) return 2;
	if (memcmp(walk.data(), e->walk[m].data(), walk.size() * 2u) != 0) return 3;
	return 0;
}
#endif

/* rows as 64-bit numbers, whatever the build's row type */
extern "C" void emu_rank4_64(void* p, int mirror, uint64_t row, uint64_t* lf, uint32_t* L)
{
	EmuIndex* e = (EmuIndex*)p;
	bt_row r[4];
	bt_rank4(e->d[mirror ? 1 : 0], (bt_row)row, r, L);
	for (int c = 0; c < 4; c++) lf[c] = r[c];
}
/* text length, the number the image's rows are offset by (wide build, BT_WIDE_ROW_BIAS), sizeof(bt_row) */
extern "C" void emu_index_dims(void* p, uint64_t out[3])
{
	EmuIndex* e = (EmuIndex*)p;
	out[0] = e->h[0].len;
#if BT_WIDE
	out[1] = e->h[0].rowBias;
#else
	out[1] = 0;
#endif
	out[2] = sizeof(bt_row);
}
/* sequence t of the index: its name into name[cap] (NUL-terminated), its length returned; -1 past the last one */
extern "C" long long emu_index_ref(void* p, uint32_t t, char* name, uint32_t cap)
{
	EmuIndex* e = (EmuIndex*)p;
	const BtIndexHost& h = e->h[0];
	if (t >= h.plen.size()) return -1;
	snprintf(name, cap, "%s", t < h.refnames.size() ? h.refnames[t].c_str() : "");
	return (long long)h.plen[t];
}
#if !BT_WIDE
extern "C" void emu_rank4(void* p, int mirror, uint32_t row, uint32_t* lf, uint32_t* L)
{
	EmuIndex* e = (EmuIndex*)p;
	bt_rank4(e->d[mirror ? 1 : 0], row, lf, L);
}
/* the same from the index files' side layout */
extern "C" void emu_rank4_sides(void* p, int mirror, uint32_t row, uint32_t* lf, uint32_t* L)
{
	EmuIndex* e = (EmuIndex*)p;
	bt_rank4_sides(e->d[mirror ? 1 : 0], row, lf, L);
}
#endif

/* Same contract as bt_align_batch (host pointers); nLanes lock-step lanes. */
template <bool RL>
static int emu_run(void* p, const bt_policy* pol, const bt_read_batch* in, bt_hit_batch* out,
                   bt_op_counts* counts, uint32_t nLanes, uint32_t frCap, uint32_t entCap, uint32_t palCap, bool lite)
{
	EmuIndex* e = (EmuIndex*)p;
	BtCold cold;
	memset(&cold, 0, sizeof(cold));
	BtProgram& P = cold.P;
	int rc = bt_host_compile_program(*pol, &P);
	if (rc != BT_OK) return rc;
	BtBatchDev& B = cold.B;
	B.seq = in->seq; B.qual = in->qual; B.len = in->len; B.seed = in->seed; B.n_reads = in->n_reads; B.stride = in->stride;
	B.hits = (BtHitRec*)out->hits; B.hit_cap = out->hit_cap; B.n_hits = out->n_hits; B.status = out->status;
	B.mm_pool = out->mm_pool; B.mm_pool_cap = out->mm_pool_cap;
	uint32_t mmUsed = 0; B.mm_pool_used = &mmUsed;
	cold.ix[0] = e->d[0]; cold.ix[1] = e->d[1];
	cold.curBid = 0; cold.ring[0] = cold.B;            /* one batch, nothing carried */
	BtHot H;
	memset(&H, 0, sizeof(H));
	BtWarm W;
	memset(&W, 0, sizeof(W));
	for (int m = 0; m < 2; m++) {
		H.blk[m] = e->d[m].blk; H.zBlk[m] = e->d[m].zBlk; H.zPos[m] = e->d[m].zPos; W.zOff[m] = e->d[m].zOff;
		W.offMask[m] = e->d[m].offMask; W.ftab[m] = e->d[m].ftab; W.offs[m] = e->d[m].offs; W.offRate[m] = e->d[m].offRate;
		W.ftabChars[m] = e->d[m].ftabChars; W.len[m] = e->d[m].len;
		W.loc[m] = e->d[m].loc; W.rtxt[m] = e->d[m].rtxt; W.walk[m] = e->d[m].walk;
#if !BT_WIDE
		if (!(getenv("EMU_JUMP_OFF") && atoi(getenv("EMU_JUMP_OFF")))) { W.jump[m] = e->d[m].jump; W.jumpMeta[m] = e->d[m].jumpMeta; W.jumpChars[m] = e->d[m].jump ? e->d[m].jumpChars : 0u; }
#endif
		for (int k = 0; k < 5; k++) H.fchr[m][k] = e->d[m].fchr[k];
#if BT_WIDE
		H.segBase[m] = e->d[m].segBase; H.segShift = e->d[m].segShift; W.rowLim[m] = e->d[m].rowLim;
#endif
	}
	W.locOn = (e->d[0].loc && (!e->mirror || e->d[1].loc) && !(getenv("EMU_LOCUS_OFF") && atoi(getenv("EMU_LOCUS_OFF")))) ? 1u : 0u;
	H.seq = in->seq; H.qual = in->qual; H.stride = in->stride; H.n_reads = in->n_reads;
	entCap = (entCap + 7u) & ~7u;
	std::vector<BtU4> frames4((size_t)nLanes * frCap * 4);
	std::vector<BtU4> pairs4((size_t)nLanes * entCap * BT_ENT_PIECES);
	std::vector<uint16_t> meta((size_t)nLanes * entCap + 8);
	std::vector<uint64_t> pals((size_t)nLanes * palCap);
	std::vector<uint32_t> tos((size_t)nLanes * BT_LDS_WORDS);
	std::vector<uint32_t> rlbuf((size_t)nLanes * BT_RL_WORDS);       /* the "LDS" copies of the reads (RL) */
	std::vector<BtLane> lanes(nLanes);
	std::vector<BtScratch> scr(nLanes);
	std::vector<BtRes> res(nLanes);
	std::vector<char> drained(nLanes, 0);
	unsigned long long CNT[CN_N];
	memset(CNT, 0, sizeof(CNT));
	BtArena arena;
	arena.frames = (uint32_t*)frames4.data(); arena.pairs = (uint32_t*)pairs4.data(); arena.meta = meta.data(); arena.pals = pals.data();
	arena.frCap = frCap; arena.entCap = entCap; arena.palCap = palCap; arena.pad = 0;
	EMU_POISON(frames4.data(), frames4.size() * sizeof(BtU4)); EMU_POISON(pairs4.data(), pairs4.size() * sizeof(BtU4));
	EMU_POISON(meta.data(), meta.size() * 2); EMU_POISON(pals.data(), pals.size() * 8);
	EMU_POISON(tos.data(), tos.size() * 4); EMU_POISON(rlbuf.data(), rlbuf.size() * 4);
	for (uint32_t g = 0; g < nLanes; g++) {
		memset(&lanes[g], 0, sizeof(BtLane));
		memset(&res[g], 0, sizeof(BtRes));
		lanes[g].state = ST_IDLE;
		scr[g].a = &arena; scr[g].slot = g;
		
		scr[g].tos = tos.data() + g; scr[g].tosStride = nLanes;
		scr[g].rl = rlbuf.data() + g;
		scr[g].tosRec = scr[g].tos + (size_t)BT_CC_WORDS * nLanes;
		scr[g].noCC = BT_WIDE ? 1u : (lite ? (BT_LITE_CC ? 2u : 1u) : 0u);      /* (the wide build has no candidate caches: a range-stack entry is 64 bytes) */
		scr[g].rlMax = lite ? BT_RL3_MAXLEN : BT_RL_MAXLEN;
	}
	uint32_t next = 0, live = nLanes;
	/* EMU_PARK_EVERY=<n>: every lane is parked and adopted again (carry-over) in one round out of n, at random */
	const uint32_t parkEvery = getenv("EMU_PARK_EVERY") ? (uint32_t)atoi(getenv("EMU_PARK_EVERY")) : 0u;
	uint32_t parkRng = 12345u;
	while (live > 0) {
		for (uint32_t g = 0; g < nLanes; g++) {
			if (drained[g]) continue;
			BtLane& L = lanes[g];
			BtReq req;
			req.tally = 0;
			for (;;) {
				if (L.state == ST_IDLE) {
					if (next >= in->n_reads) { drained[g] = 1; live--; break; }
					bt_lane_start<RL>(L, P, H, cold, scr[g], next++);
				}
				bt_lane_run<RL>(L, P, H, W, cold, scr[g], res[g], req, CNT);
				if (L.state != ST_IDLE) break;
			}
			CNT[CN_TLFEX] += req.tally >> 16; CNT[CN_TLF1] += req.tally & 0xffffu;      /* locus mode's steps decided by the text (also a finished read's) */
			if (drained[g]) continue;
			if (req.kind == RQ_FETCH && (L.state == ST_LOC_REC || (L.state == ST_RESOLVE_DONE && W.locOn))) BT_COUNT(CN_LOCREC);
			if (req.kind == RQ_FETCH && L.state == ST_LOC_TXT) BT_COUNT(CN_TXTWIN);
			BT_COUNT(CN_ITERS);
			L.iters++;
			if (parkEvery && (parkRng = parkRng * 1664525u + 1013904223u, (parkRng >> 8) % parkEvery == 0)) {
				/* carry-over as the kernel does it (bt_kernels.hip): the lane's state and pending request survive, what sat
				 * in LDS does not -- the top-of-stack record and candidate cache are invalid, the read is loaded again */
				for (uint32_t k = 0; k < BT_LDS_WORDS; k++) tos[(size_t)k * nLanes + g] = 0xdeadbeefu;
				for (uint32_t k = 0; k < BT_RL_WORDS; k++) rlbuf[(size_t)k * nLanes + g] = 0xdeadbeefu;
				{
					/* ... and the lane's state goes through the pool record's bytes as the kernel copies it: in builds that keep
					 * the read in LDS only what lies before the register window (offsetof cs0), the rest starts from zero */
					static_assert((sizeof(BtLane) + 15) / 16 <= 17, "a parked lane fits the pool record (bt_kernels.h: BT_POOL_REQ)");
					uint8_t rec[sizeof(BtLane)];
					const size_t keep = RL ? offsetof(BtLane, cs0) : sizeof(BtLane);
					memcpy(rec, &L, keep);
					BtLane fresh;
					memset(&fresh, 0, sizeof(fresh));
					memcpy(&fresh, rec, keep);
					L = fresh;
				}
				L.tosValid = 0; L.ccValid = 0;
				if (RL) for (uint32_t base = 0; base < L.plen; base += 16u)
					bt_rl_store_chunk(scr[g], base, bt_ld4(in->seq + L.roff + base), bt_ld4(in->qual + L.roff + base));
			}
			if (req.kind == RQ_RANK) {
				if (L.lfk == LFK_CHASE) BT_COUNT(CN_CHASE);
				else if (L.lfk == LFK_EX2) BT_COUNT(CN_LFEX);
				else if (L.lfk == LFK_C2) BT_COUNT(CN_LF2);
				else BT_COUNT(CN_LF1);
				if (req.n == 2 && (bt_row)req.a / 448u == (bt_row)req.x / 448u) BT_COUNT(CN_SAMEPAIR);
				const BtIndexDev& ix = e->d[L.mirror];
#if BT_WIDE
				{
					/* as the kernel lays a rank answer out in the wide build (BtRes): quartets of 64-bit rows, two pieces each */
					bt_row lf[4]; uint32_t la, dummy;
					EMU_POISON(&res[g], sizeof(BtRes));
					memset(&res[g].x, 0, 16);
					bt_rank4(ix, (bt_row)req.a, lf, &la);
					memcpy(&res[g].q[0], lf, 32);
					res[g].x.x = la;
					if (req.n == 2) { bt_rank4(ix, (bt_row)req.x, lf, &dummy); memcpy(&res[g].q[2], lf, 32); }
					continue;
				}
#else
				uint32_t lf[4], la, dummy;
				EMU_POISON(&res[g].q[1], 16); EMU_POISON(&res[g].q[2], 16);     /* what the kernel's rank branch leaves unset */
				memset(&res[g].q[3], 0, 16); memset(&res[g].x, 0, 16);
				bt_rank4(ix, (uint32_t)req.a, lf, &la);
				res[g].q[0].x = lf[0]; res[g].q[0].y = lf[1]; res[g].q[0].z = lf[2]; res[g].q[0].w = lf[3];
				res[g].q[2].x = la;
				if (req.wchunk != 0xffffu) {
					memcpy(&res[g].q[3], in->seq + L.roff + (size_t)req.wchunk * 16, 16);
					memcpy(&res[g].x, in->qual + L.roff + (size_t)req.wchunk * 16, 16);
				}
				if (req.n == 2) {
					bt_rank4(ix, (uint32_t)req.x, lf, &dummy);
					res[g].q[1].x = lf[0]; res[g].q[1].y = lf[1]; res[g].q[1].z = lf[2]; res[g].q[1].w = lf[3];
				}
#endif
			} else {
				memset(&res[g], 0, sizeof(BtRes));
				for (uint32_t k = 0; k < req.n; k++) memcpy(&res[g].q[k], (const uint8_t*)(uintptr_t)req.a + 16 * k, 16);
				if (req.x) memcpy(&res[g].x, (const void*)(uintptr_t)req.x, 16);
			}
		}
	}
	out->mm_pool_used = mmUsed < out->mm_pool_cap ? mmUsed : out->mm_pool_cap;
	if (getenv("BT_EMU_VERBOSE")) fprintf(stderr, "[emu] jump table: %llu look-ups, %llu two-row + %llu one-row steps behind them; ftab look-ups in all %llu\n",
	                                      CNT[CN_JUMPS], CNT[CN_JLF2], CNT[CN_JLF1], CNT[CN_FTAB]);
	if (counts) {
		counts->lfex = CNT[CN_LFEX]; counts->lf2 = CNT[CN_LF2]; counts->lf1 = CNT[CN_LF1]; counts->chase = CNT[CN_CHASE];
		counts->ftab = CNT[CN_FTAB]; counts->offs = CNT[CN_OFFS]; counts->rstarts = CNT[CN_RSTARTS];
		counts->frames = CNT[CN_FRAMES]; counts->lane_iters = CNT[CN_ITERS]; counts->same_pair = CNT[CN_SAMEPAIR];
		counts->rescans = CNT[CN_RESCAN]; counts->cand_scans = CNT[CN_CANDSCAN]; counts->fetches = CNT[CN_FETCH];
		/* what locus mode decided by the text is part of the reference's op counts all the same (bt_op_counts) */
		counts->loc_lfex = CNT[CN_TLFEX]; counts->loc_lf1 = CNT[CN_TLF1]; counts->loc_chase = CNT[CN_TCHASE];
		counts->loc_records = CNT[CN_LOCREC]; counts->loc_windows = CNT[CN_TXTWIN];
		counts->lfex += CNT[CN_TLFEX]; counts->same_pair += CNT[CN_TLFEX]; counts->lf1 += CNT[CN_TLF1]; counts->chase += CNT[CN_TCHASE];
		/* ... and so is what lies behind the jump table's look-ups */
		counts->lf2 += CNT[CN_JLF2]; counts->lf1 += CNT[CN_JLF1]; counts->same_pair += CNT[CN_JSAME];
	}
	return BT_OK;
}


/* bt_best_kernel's loop -- the wavefront automaton of bt_best.h -- for one "wavefront" of W lanes gone through side by side:
 * the same decisions as the kernel's (hot round or cold sweep, new reads, the gate of the ended streaks), its ballots
 * being counts over the lanes.  Checks what the loop has to get right on top of the pieces it calls: every read run
 * exactly once, by some lane, each lane's pieces in the order bf_run_read / bf_run_pair go through them, and the loop ends. */
static void emu_best_wave(std::vector<BfLane>& XS, const BtBatchDev& B, uint32_t n, uint32_t kind,
                          uint32_t coldMin, uint32_t takeMin, uint32_t sendPeriod, uint32_t sendMin)
{
	const size_t W = XS.size();
	const bool sweepTwice = getenv("BT_BEST_SWEEP_TWICE") && atoi(getenv("BT_BEST_SWEEP_TWICE")) != 0;
	std::vector<BfAuto> S(W);
	std::vector<BfLeafSt> leafs(W);       /* (the kernel keeps these in LDS: BfAuto::leafp) */
	for (size_t l = 0; l < W; l++) { BfAuto& a = S[l]; memset(&a, 0, sizeof(a)); memset(&leafs[l], 0, sizeof(BfLeafSt)); a.leafp = &leafs[l]; a.phase = BA_TAKE; a.kind = kind; }
	uint32_t next = 0, round = 0, sweeps = 0;
	/* the wave model's tallies (BT_EMU_VERBOSE): rounds and lanes per piece */
	unsigned long long stepR = 0, stepL = 0, sendR = 0, sendL = 0, chaseR = 0, chaseL = 0, sweepL = 0, sweep2 = 0, sweep2L = 0, takeS = 0, takeL = 0, idleL = 0;
	auto take = [&]() -> uint32_t { return next < n ? next++ : 0xffffffffu; };
	for (;;) {
		uint32_t nHot = 0, nCold = 0, nTake = 0, nSend = 0;
		for (size_t l = 0; l < W; l++) {
			if (BA_IS_HOT(S[l].phase)) nHot++; else if (S[l].phase != BA_IDLE) nCold++;
			if (S[l].phase == BA_TAKE || S[l].phase == BA_END) nTake++;
			if (S[l].phase == BA_SEND) nSend++;
		}
		if (!nHot && !nCold) break;
		const bool takeOk = nTake >= takeMin || nTake == nHot + nCold;
		const uint32_t nSweep = takeOk ? nCold : nCold - nTake;
		if (nHot && nSweep < coldMin) {
			round++;
			const bool sendOk = (round % sendPeriod) == 0u || nSend >= sendMin;
			uint32_t a = 0, b = 0, c = 0;
			for (size_t l = 0; l < W; l++) if (BA_IS_HOT(S[l].phase)) {
				const uint32_t did = bf_auto_hot(XS[l], S[l], sendOk);
				a += did & 1u; b += (did >> 1) & 1u; c += (did >> 2) & 1u;
				if (!did) idleL++;
			}
			idleL += nCold;
			if (a) { stepR++; stepL += a; } if (b) { sendR++; sendL += b; } if (c) { chaseR++; chaseL += c; }
			continue;
		}
		uint32_t t = 0, k = 0;
		for (size_t l = 0; l < W; l++) if (!BA_IS_HOT(S[l].phase)) {
			const bool wasTake = S[l].phase == BA_TAKE || S[l].phase == BA_END;
			if (S[l].phase != BA_IDLE && (takeOk || !wasTake)) k++;
			bf_auto_cold(XS[l], B, S[l], takeOk, take);
			if (wasTake && takeOk && S[l].phase != BA_IDLE && S[l].phase != BA_TAKE) t++;
		}
		sweepL += k;
		if (t) { takeS++; takeL += t; }
		k = 0;
		if (sweepTwice) for (size_t l = 0; l < W; l++) if (BA_IS_PENDING(S[l].phase)) { bf_auto_cold(XS[l], B, S[l], false, take); k++; }
		if (k) { sweep2++; sweep2L += k; }
		sweeps++;
	}
	if (getenv("BT_EMU_VERBOSE"))
		fprintf(stderr, "[emu] automaton: %u reads, %zu lanes, gates %u/%u/%u/%u: %u hot rounds (step %llu x %.1f lanes, send %llu x %.1f, chase %llu x %.1f; idle lane-rounds %llu), "
		        "%u cold sweeps x %.1f lanes (second pass %llu x %.1f; reads taken in %llu x %.1f)\n", n, W, coldMin, takeMin, sendPeriod, sendMin, round,
		        stepR, stepR ? (double)stepL / stepR : 0.0, sendR, sendR ? (double)sendL / sendR : 0.0, chaseR, chaseR ? (double)chaseL / chaseR : 0.0, idleL,
		        sweeps, sweeps ? (double)sweepL / sweeps : 0.0, sweep2, sweep2 ? (double)sweep2L / sweep2 : 0.0, takeS, takeS ? (double)takeL / takeS : 0.0);
}
static uint32_t emu_env_u32(const char* name, uint32_t dflt) { const char* v = getenv(name); return v && *v ? (uint32_t)strtoul(v, nullptr, 0) : dflt; }

/* The best-first engine (bowtie_amd/csrc/bt_best.h): the kernel's loop with 24 lanes side by side, each in an arena of
 * arenaWords 32-bit words (BT_EMU_BEST_NESTED=1: one read after the other by bf_run_read, as bt_best_nested_kernel does). */
static int emu_run_best(void* p, const bt_policy* pol, const bt_read_batch* in, bt_hit_batch* out,
                        bt_op_counts* counts, uint32_t arenaWords)
{
	EmuIndex* e = (EmuIndex*)p;
	BfProgram P;
	int rc = bt_host_compile_best(*pol, &P);
	if (rc != BT_OK) return rc;
	if (P.needMirror && !e->mirror) return BT_ERR_ARG;
	BtBatchDev B;
	memset(&B, 0, sizeof(B));
	B.seq = in->seq; B.qual = in->qual; B.len = in->len; B.seed = in->seed; B.n_reads = in->n_reads; B.stride = in->stride;
	B.hits = (BtHitRec*)out->hits; B.hit_cap = out->hit_cap; B.n_hits = out->n_hits; B.status = out->status;
	B.mm_pool = out->mm_pool; B.mm_pool_cap = out->mm_pool_cap;
	uint32_t mmUsed = 0; B.mm_pool_used = &mmUsed;
	BfLane X;
	if (emu_env_u32("BT_EMU_BEST_NESTED", 0)) {
		std::vector<uint32_t> arena(arenaWords);
		memset(&X, 0, sizeof(X));
		X.A = arena.data(); X.cap = arenaWords; X.ix = e->d; X.P = &P;
		for (uint32_t rd = 0; rd < in->n_reads; rd++) bf_run_read(X, B, rd);
	} else {
		/* arenas from calloc: untouched pages cost nothing */
		const size_t WL = emu_env_u32("BT_EMU_WAVE_LANES", 24), W = in->n_reads < WL ? (in->n_reads ? in->n_reads : 1u) : WL;
		uint32_t* arenas = (uint32_t*)calloc(W * (size_t)arenaWords + 1u, 4);
		if (!arenas) return BT_ERR_DEVICE;
		std::vector<BfLane> XS(W);
		for (size_t l = 0; l < W; l++) { memset(&XS[l], 0, sizeof(BfLane)); XS[l].A = arenas + l * (size_t)arenaWords; XS[l].cap = arenaWords; XS[l].ix = e->d; XS[l].P = &P; }
		emu_best_wave(XS, B, in->n_reads, 1u, emu_env_u32("BT_BEST_COLD_MIN", 6), emu_env_u32("BT_BEST_TAKE_MIN", 5),
		              emu_env_u32("BT_BEST_SEND_PERIOD", 3), emu_env_u32("BT_BEST_SEND_MIN", 4));
		X = XS[0];
		for (size_t l = 1; l < W; l++) {
			X.c_lfex += XS[l].c_lfex; X.c_lf2 += XS[l].c_lf2; X.c_lf1 += XS[l].c_lf1; X.c_chase += XS[l].c_chase; X.c_ftab += XS[l].c_ftab;
			X.c_offs += XS[l].c_offs; X.c_rst += XS[l].c_rst; X.c_same += XS[l].c_same; X.c_frames += XS[l].c_frames;
		}
		free(arenas);
	}
	out->mm_pool_used = mmUsed < out->mm_pool_cap ? mmUsed : out->mm_pool_cap;
	if (counts) {
		counts->lfex = X.c_lfex; counts->lf2 = X.c_lf2; counts->lf1 = X.c_lf1; counts->chase = X.c_chase;
		counts->ftab = X.c_ftab; counts->offs = X.c_offs; counts->rstarts = X.c_rst; counts->same_pair = X.c_same;
		counts->frames = X.c_frames;
	}
	return BT_OK;
}

/* Paired-end through the best-first engine (bf_run_pair).  Hits come in pairs (upstream mate, downstream
 * mate) in the hit_cap slots of each pair. */
extern "C" int emu_align_pairs(void* p, const bt_policy* pol, const bt_read_batch* in1, const bt_read_batch* in2,
                               bt_hit_batch* out, bt_op_counts* counts, uint32_t arenaWords)
{
	EmuIndex* e = (EmuIndex*)p;
	if (in1->n_reads != in2->n_reads) return BT_ERR_ARG;
	BfProgram P;
	int rc = bt_host_compile_best_paired(*pol, &P);
	if (rc != BT_OK) return rc;
	if (P.needMirror && !e->mirror) return BT_ERR_ARG;
	if (!e->haveRef) {
		rc = bt_host_ref_load(e->base, e->h[0], &e->ref);
		if (rc != BT_OK) return rc;
		e->refd.bits = e->ref.bits.data(); e->refd.nmask = e->ref.nmask.data(); e->refd.start = e->ref.start.data();
		e->refd.approxLen = e->ref.approxLen.data(); e->refd.nRefs = (uint32_t)e->ref.start.size(); e->refd.pad = 0;
		e->haveRef = true;
	}
	BtBatchDev B;
	memset(&B, 0, sizeof(B));
	B.seq = in1->seq; B.qual = in1->qual; B.len = in1->len; B.seed = in1->seed; B.n_reads = in1->n_reads; B.stride = in1->stride;
	B.seq2 = in2->seq; B.qual2 = in2->qual; B.len2 = in2->len; B.seed2 = in2->seed; B.stride2 = in2->stride;
	B.hits = (BtHitRec*)out->hits; B.hit_cap = out->hit_cap; B.n_hits = out->n_hits; B.status = out->status;
	B.mm_pool = out->mm_pool; B.mm_pool_cap = out->mm_pool_cap;
	uint32_t mmUsed = 0; B.mm_pool_used = &mmUsed;
	if (arenaWords < 256u) arenaWords = 1u << 22;
	BfLane X;
	if (BF_IS_V1(P) || emu_env_u32("BT_EMU_BEST_NESTED", 0)) {
		std::vector<uint32_t> arena(arenaWords);
		memset(&X, 0, sizeof(X));
		X.A = arena.data(); X.cap = arenaWords; X.ix = e->d; X.P = &P; X.ref = &e->refd;
		for (uint32_t rd = 0; rd < in1->n_reads; rd++) { if (BF_IS_V1(P)) bf_run_pair_v1(X, B, rd); else bf_run_pair(X, B, rd); }
	} else {
		const size_t WL = emu_env_u32("BT_EMU_WAVE_LANES", 24), W = in1->n_reads < WL ? (in1->n_reads ? in1->n_reads : 1u) : WL;
		uint32_t* arenas = (uint32_t*)calloc(W * (size_t)arenaWords + 1u, 4);
		if (!arenas) return BT_ERR_DEVICE;
		std::vector<BfLane> XS(W);
		for (size_t l = 0; l < W; l++) { memset(&XS[l], 0, sizeof(BfLane)); XS[l].A = arenas + l * (size_t)arenaWords; XS[l].cap = arenaWords; XS[l].ix = e->d; XS[l].P = &P; XS[l].ref = &e->refd; }
		emu_best_wave(XS, B, in1->n_reads, 2u, emu_env_u32("BT_BEST_COLD_MIN", 6), emu_env_u32("BT_BEST_TAKE_MIN", 5),
		              emu_env_u32("BT_BEST_SEND_PERIOD", 3), emu_env_u32("BT_BEST_SEND_MIN", 4));
		X = XS[0];
		for (size_t l = 1; l < W; l++) {
			X.c_lfex += XS[l].c_lfex; X.c_lf2 += XS[l].c_lf2; X.c_lf1 += XS[l].c_lf1; X.c_chase += XS[l].c_chase; X.c_ftab += XS[l].c_ftab;
			X.c_offs += XS[l].c_offs; X.c_rst += XS[l].c_rst; X.c_same += XS[l].c_same; X.c_frames += XS[l].c_frames;
		}
		free(arenas);
	}
	out->mm_pool_used = mmUsed < out->mm_pool_cap ? mmUsed : out->mm_pool_cap;
	if (counts) {
		counts->lfex = X.c_lfex; counts->lf2 = X.c_lf2; counts->lf1 = X.c_lf1; counts->chase = X.c_chase;
		counts->ftab = X.c_ftab; counts->offs = X.c_offs; counts->rstarts = X.c_rst; counts->same_pair = X.c_same;
		counts->frames = X.c_frames;
	}
	return BT_OK;
}

/* rl_mode: 0 = as the kernel launcher decides (reads of <= BT_RL_MAXLEN bases keep their read in "LDS"),
 * 1 = force the register-window build of the automaton, 2 = the lite layout of the 3-waves build */
extern "C" int emu_align_batch(void* p, const bt_policy* pol, const bt_read_batch* in, bt_hit_batch* out,
                               bt_op_counts* counts, uint32_t nLanes, uint32_t frCap, uint32_t entCap, uint32_t palCap,
                               uint32_t rl_mode)
{
	uint32_t maxLen = 0;
	for (uint32_t i = 0; i < in->n_reads; i++) if (in->len[i] > maxLen) maxLen = in->len[i];
	/* the stateful best-first workers: entCap doubles as the arena size in words (0 = 4 M words) */
	if (pol->best) return emu_run_best(p, pol, in, out, counts, entCap >= 256u ? entCap : (1u << 22));
	/* rl_mode 2 = the 3-waves-per-SIMD layout: read in LDS (<= 104 bases), no candidate caches */
	if (rl_mode == 2 && maxLen <= BT_RL3_MAXLEN) return emu_run<true>(p, pol, in, out, counts, nLanes, frCap, entCap, palCap, true);
	if (rl_mode == 0 && maxLen <= BT_RL_MAXLEN) return emu_run<true>(p, pol, in, out, counts, nLanes, frCap, entCap, palCap, false);
	return emu_run<false>(p, pol, in, out, counts, nLanes, frCap, entCap, palCap, false);
}
