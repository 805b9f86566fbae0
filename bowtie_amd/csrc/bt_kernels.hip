/*
 * bt_kernels.hip -- gfx950 kernels of the FM-index search path.
 *
 *   bt_search_kernel : one lane = one read automaton (bt_core.h).  The 64 reads of a wavefront
 *                      advance in lock step from request to request; the rank gathers of all lanes
 *                      (one 32-byte rank block per BWT row -- four absolute counters and the two
 *                      bit planes of the block's 64 rows, bt_rank.h -- or up to 4+1 16-byte pieces
 *                      of a fetch) are issued together at one program point, whatever phase /
 *                      backtrack frame / SA walk each read is in.  Lanes that finish a read pull the
 *                      next one from a global counter, so wavefronts stay full until the batch drains.
 *   bt_probe_*       : known-answer probes of rank/LF and the SA walk.
 *
 * Replaces (reference, CPU): the worker loops of ebwt_search.cpp:1130/1606/2056/2378 and
 * everything they call for a read.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stddef.h>
#include "bt_core.h"
#include "bt_kernels.h"

/* ---- deriving the rank blocks from the index files' side layout (bt_rank.h), once per index at load -------------
 * one wavefront per block of 64 BWT rows: every lane reads its row's symbol, two ballots make the bit planes, lane 0
 * ranks the block's first row the old way for the four absolute counters */
#if !BT_WIDE
__device__ __forceinline__ uint32_t dev_rowL_sides(const BtIndexDev& ix, uint32_t row)
{
	const uint32_t sideNum = row / BT_SIDE_SYMS, charOff = row - sideNum * BT_SIDE_SYMS;
	const uint32_t li = (sideNum & 1u) ? charOff : (BT_SIDE_SYMS - 1u - charOff);
	return ((uint32_t)BT_GP(const uint8_t, ix.ebwt)[(uint64_t)sideNum * 64u + (li >> 2)] >> (2u * (li & 3u))) & 3u;
}
__global__ __launch_bounds__(256) void bt_blk_build_kernel(BtIndexDev ix, uint8_t* out, uint32_t nBlocks)
{
	const uint32_t b = blockIdx.x * 4u + threadIdx.x / 64u, lane = threadIdx.x & 63u;
	if (b >= nBlocks) return;
	const uint64_t row = (uint64_t)b * BT_BLK_ROWS + lane;
	const uint32_t L = row <= ix.len ? dev_rowL_sides(ix, (uint32_t)row) : 0u;
	const unsigned long long p0 = __ballot(L & 1u), p1 = __ballot(L & 2u);
	if (lane == 0) {
		uint32_t lf[4] = {0, 0, 0, 0}, dummy;
		if (row <= (uint64_t)ix.len + 1u) bt_rank4_sides(ix, (uint32_t)row, lf, &dummy);
		BtU4 o, p;
		o.x = lf[0]; o.y = lf[1]; o.z = lf[2]; o.w = lf[3];
		p.x = (uint32_t)p0; p.y = (uint32_t)(p0 >> 32); p.z = (uint32_t)p1; p.w = (uint32_t)(p1 >> 32);
		bt_st4(out + (uint64_t)b * BT_BLK_BYTES, o); bt_st4(out + (uint64_t)b * BT_BLK_BYTES + 16, p);
	}
}
extern "C" int bt_launch_blk_build(const BtIndexDev* ix, uint8_t* out, uint32_t nBlocks, void* stream)
{
	hipLaunchKernelGGL(bt_blk_build_kernel, dim3((nBlocks + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, *ix, out, nBlocks);
	return (int)hipGetLastError();
}

/* ---- the locus image (bt_rank.h), derived once per index at load (what bt_loc_build_host does on the host) ------------------
 * pass 1, one lane per BWT row: walk to a sampled row as Ebwt::reportChaseOne does (ebwt.h:2727-2746) -> SA[row] into the
 * row's record, the walk's length into walk[SA] (by text offset), and the row's own BWT character -- the text's base to the
 * left of its suffix -- into the reversed text.  pass 2, one lane per row: the 48 characters to the left of the row's suffix,
 * out of the reversed text, into the record. */
__global__ __launch_bounds__(256) void bt_loc_sa_kernel(BtIndexDev ix, BtU4* loc, uint32_t* rtxt, uint16_t* walk)
{
	const uint64_t nRows = (uint64_t)ix.len + 1u;
	for (uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r0 < nRows; r0 += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t row = (uint32_t)r0, j = 0, first = 0;
		while ((row & ix.offMask) != row && row != ix.zOff) {
			uint32_t lf[4], L;
			bt_rank4(ix, row, lf, &L);
			if (j == 0) first = L;
			row = lf[L];
			j++;
		}
		if (j == 0 && (uint32_t)r0 != ix.zOff) { uint32_t lf[4]; bt_rank4(ix, (uint32_t)r0, lf, &first); }
		const uint32_t sa = (row == ix.zOff ? 0u : BT_GP(const uint32_t, ix.offs)[row >> ix.offRate]) + j;
		BT_GP(uint32_t, loc)[r0 * 4u] = sa;
		BT_GP(uint16_t, walk)[sa] = (uint16_t)(j < 0xffffu ? j : 0xffffu);
		if (sa > 0) {
			const uint32_t y = ix.len - sa;                 /* T[sa-1] is base len-1-(sa-1) of the reversed text */
			atomicOr(rtxt + (y >> 4), first << (2u * (y & 15u)));
		}
	}
}
/* pass 1 again, a chain at a time (round 6; bt_rank.h: bt_loc_chain): one lane per sampled row */
__global__ __launch_bounds__(256) void bt_loc_chain_kernel(BtIndexDev ix, BtU4* loc, uint32_t* rtxt, uint16_t* walk)
{
	const uint64_t nChains = bt_loc_chain_count(ix);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nChains; i += (uint64_t)gridDim.x * blockDim.x)
		bt_loc_chain(ix, i, loc, rtxt, walk);
}
__global__ __launch_bounds__(256) void bt_loc_ctx_kernel(BtIndexDev ix, BtU4* loc, const uint32_t* rtxt)
{
	const uint64_t nRows = (uint64_t)ix.len + 1u;
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nRows; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t sa = BT_GP(const uint32_t, loc)[r * 4u];
		const uint32_t y0 = ix.len - sa, w = y0 >> 4, s = 2u * (y0 & 15u);
		uint32_t t[4], c[3];
		BT_UNROLL
		for (int k = 0; k < 4; k++) t[k] = BT_GP(const uint32_t, rtxt)[w + k];
		BT_UNROLL
		for (int k = 0; k < 3; k++) c[k] = (uint32_t)((((uint64_t)t[k + 1] << 32) | t[k]) >> s);
		BtU4 v; v.x = sa; v.y = c[0]; v.z = c[1]; v.w = c[2];
		bt_st4(loc + r, v);
	}
}
/* loc: len + 1 records; rtxtAlloc: bt_rtxt_words(len) words, zeroed by the caller; walk: len + 1 entries */
extern "C" int bt_launch_loc_build(const BtIndexDev* ix, BtU4* loc, uint32_t* rtxtAlloc, uint16_t* walk, void* stream)
{
	hipDeviceProp_t prop; int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
	const uint64_t nRows = (uint64_t)ix->len + 1u;
	uint64_t nb = (nRows + 255u) / 256u;
	const uint64_t cap = (uint64_t)prop.multiProcessorCount * 64u;
	if (nb > cap) nb = cap;
	/* BT_LOC_BUILD=rows: round 5's pass 1 (a walk per row) instead of a chain per sampled row */
	const char* how = getenv("BT_LOC_BUILD");
	if (how && !strcmp(how, "rows")) hipLaunchKernelGGL(bt_loc_sa_kernel, dim3((uint32_t)nb), dim3(256), 0, (hipStream_t)stream, *ix, loc, rtxtAlloc + BT_RTXT_PAD_WORDS, walk);
	else {
		const uint64_t nChains = bt_loc_chain_count(*ix);
		uint64_t ncb = (nChains + 255u) / 256u;
		if (ncb > cap) ncb = cap;
		hipLaunchKernelGGL(bt_loc_chain_kernel, dim3((uint32_t)ncb), dim3(256), 0, (hipStream_t)stream, *ix, loc, rtxtAlloc + BT_RTXT_PAD_WORDS, walk);
	}
	hipLaunchKernelGGL(bt_loc_ctx_kernel, dim3((uint32_t)nb), dim3(256), 0, (hipStream_t)stream, *ix, loc, rtxtAlloc + BT_RTXT_PAD_WORDS);
	return (int)hipGetLastError();
}

/* the jump table (bt_rank.h: BtIndexDev::jump), an entry per lane */
__global__ __launch_bounds__(256) void bt_jump_build_kernel(BtIndexDev ix, uint32_t K, uint32_t* jump, uint16_t* meta)
{
	const uint64_t n = 1ull << (2u * K);
	for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t top, bot, m;
		bt_jump_entry(ix, (uint32_t)x, K, &top, &bot, &m);
		BT_GP(uint32_t, jump)[2u * x] = top; BT_GP(uint32_t, jump)[2u * x + 1u] = bot;
		BT_GP(uint16_t, meta)[x] = (uint16_t)m;
	}
}
extern "C" int bt_launch_jump_build(const BtIndexDev* ix, uint32_t K, uint32_t* jump, uint16_t* meta, void* stream)
{
	hipDeviceProp_t prop; int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
	uint64_t nb = ((1ull << (2u * K)) + 255u) / 256u;
	const uint64_t cap = (uint64_t)prop.multiProcessorCount * 64u;
	if (nb > cap) nb = cap;
	hipLaunchKernelGGL(bt_jump_build_kernel, dim3((uint32_t)nb), dim3(256), 0, (hipStream_t)stream, *ix, K, jump, meta);
	return (int)hipGetLastError();
}

#else
extern "C" int bt_launch_jump_build(const BtIndexDev*, uint32_t, uint32_t*, uint16_t*, void*) { return -1; }
/* the wide build: its loader derives the rank blocks on the host, straight from the file's BWT (bt_host.cpp), and it has no
 * locus image */
extern "C" int bt_launch_blk_build(const BtIndexDev*, uint8_t*, uint32_t, void*) { return -1; }
extern "C" int bt_launch_loc_build(const BtIndexDev*, BtU4*, uint32_t*, uint16_t*, void*) { return -1; }
#endif

/* EXT = true compiles in carry-over (parking at the end of a launch, adoption at the start of the next) and the
 * pick-up list of the overflow second pass; a launch that needs neither uses the leaner EXT = false build. */
/* LITE (with RL): the LDS diet that lets three blocks share a CU -- the read in 39 words (<= 104 bases) and
 * the top-of-stack record only, no candidate caches (51 KB per block instead of 72). */
template <int OCC, bool EXT, bool RL, bool LITE>
__global__ __launch_bounds__(BT_BLOCK, OCC) void bt_search_kernel(BtKernelArgs A)
{
	__shared__ uint32_t RLB[RL ? (LITE ? BT_RL3_WORDS : BT_RL_WORDS) * BT_BLOCK : 1];  /* RL: every lane's whole read */
	__shared__ unsigned long long CNT[CN_N + PS_N];
	__shared__ uint32_t TOS[(LITE ? (BT_LITE_CC ? BT_LITE_LDS_WORDS : BT_TOS_WORDS) : BT_LDS_WORDS) * BT_BLOCK];   /* per lane: candidate, top-of-stack record, its candidate (LITE: not the last) */
	__shared__ BtProgram PROG;                                 /* the phase program, read on every phase change */
	__shared__ BtWarm WARM;                                    /* index geometry (see BtWarm) */
	__shared__ BtArena ARENA;                                  /* scratch arena bases + capacities */
	/* three blocks of the LITE build share a CU's 160 KB of LDS, which gfx950 hands out in 1 280-byte granules: 42 of the
	 * CU's 128 per block, not a byte more (round 6: one more word per lane in TOS made it two blocks per CU, 15.7 -> 12.0 M reads/s) */
	static_assert(!LITE || sizeof(RLB) + sizeof(CNT) + sizeof(TOS) + sizeof(PROG) + sizeof(WARM) + sizeof(ARENA) + 64 <= 42u * 1280u,
	              "the three-block build's LDS must fit 42 granules of 1280 bytes");
	if (A.gate) { const uint32_t v = *BT_GP(const uint32_t, A.gate); if (v < A.gateLo || v > A.gateHi) return; }
	if (threadIdx.x < CN_N + PS_N) CNT[threadIdx.x] = 0;
	for (uint32_t i = threadIdx.x; i < sizeof(BtProgram) / 4; i += blockDim.x)
		((uint32_t*)&PROG)[i] = ((const uint32_t*)&A.cold->P)[i];
	for (uint32_t i = threadIdx.x; i < sizeof(BtWarm) / 4; i += blockDim.x)
		((uint32_t*)&WARM)[i] = ((const uint32_t*)A.warm)[i];
	__syncthreads();

	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	BtScratch S;
	if (threadIdx.x == 0) {
		ARENA.frames = A.frames; ARENA.pairs = A.pairs; ARENA.meta = A.meta; ARENA.pals = A.pals;
		ARENA.frCap = A.frCap; ARENA.entCap = A.entCap; ARENA.palCap = A.palCap; ARENA.pad = 0;
	}
	__syncthreads();
	S.a = &ARENA; S.slot = g;
	/* pool record: the lane's state in 16-byte pieces from the front, request at pieces 13 and 14, stamp at 15.  Builds that
	 * keep the read in LDS (RL) have no use for the register window at the end of BtLane: it is neither parked nor adopted,
	 * so it occupies no registers in them */
	constexpr int LANE_PIECES = (int)((sizeof(BtLane) + 15) / 16), LANE_PIECES_RL = (int)((offsetof(BtLane, cs0) + 15) / 16);
	constexpr int PIECES = RL ? LANE_PIECES_RL : LANE_PIECES;
	static_assert(LANE_PIECES <= BT_POOL_REQ, "pool record layout: at most BT_POOL_REQ pieces of lane state, then request and stamp");
	S.tos = TOS + threadIdx.x; S.tosStride = BT_BLOCK;
	S.tosRec = (LITE && !BT_LITE_CC) ? S.tos : S.tos + BT_CC_WORDS * BT_BLOCK;
	S.noCC = LITE ? (BT_LITE_CC ? 2u : 1u) : 0u;
	S.rlMax = LITE ? BT_RL3_MAXLEN : BT_RL_MAXLEN;
	S.rl = RLB + (RL ? threadIdx.x : 0u);

	BtLane L = {};
	L.state = ST_IDLE;
	BtReq req;
	req.kind = RQ_NONE; req.n = 0; req.a = 0; req.x = 0; req.wchunk = 0xffffu;
	bool drained = false, parkNow = false;
	const BtCold* cold = A.cold;
	uint32_t nReads = A.H.n_reads;
	if (EXT && A.orderCount) { const uint32_t v = *BT_GP(const uint32_t, A.orderCount); nReads = v < A.orderCap ? v : A.orderCap; }
#ifdef BT_PROFILE
	bool pf_prevSingle = false;
#endif
	uint32_t sc_iters = 0, sc_rounds = 0, sc_fetch = 0, sc_chase = 0, sc_lfex = 0, sc_lf2 = 0, sc_lf1 = 0, sc_same = 0, sc_locrec = 0, sc_txtwin = 0;
	uint32_t tl_acc = 0;       /* this lane's locus-mode tallies (BtReq::tally: mapLFEx | mapLF1 steps decided by the text), flushed before a half can overflow */

	if (EXT && A.pool && A.adopt) {
		/* carry-over: what lane g of the previous launch parked -- state and pending request; the scratch slot is g
		 * as ever, and the read goes back into LDS from its own batch's arrays */
		const BtPoolRec* r = A.pool + g;
		if (BT_GP(const uint32_t, r->w)[BT_POOL_STAMP_WORD] == A.launchSeq - 1u) {
			/* word by word into the lane state: a 16-byte-piece copy makes the compiler keep BtLane as twelve
			 * 4-word vectors for the whole round loop, a different (and larger) kernel than the plain build */
			uint32_t lw[4 * LANE_PIECES] = {};
			BT_UNROLL
			for (int k = 0; k < PIECES; k++) { const BtU4 v = bt_ld4((const uint8_t*)r->w + 16 * k); lw[4 * k] = v.x; lw[4 * k + 1] = v.y; lw[4 * k + 2] = v.z; lw[4 * k + 3] = v.w; }
			__builtin_memcpy(&L, lw, RL ? offsetof(BtLane, cs0) : sizeof(L));
			{ const BtU4 v = bt_ld4((const uint8_t*)r->w + 16 * BT_POOL_REQ); req.kind = v.x; req.n = v.y; req.wchunk = v.z; }
			{ const BtU4 v = bt_ld4((const uint8_t*)r->w + 16 * (BT_POOL_REQ + 1)); req.a = ((uint64_t)v.y << 32) | v.x; req.x = ((uint64_t)v.w << 32) | v.z; }
			L.tosValid = 0; L.ccValid = 0;
			if (RL) {
				const BtBatchDev* pb = &cold->ring[L.bid];
				const uint8_t* ps = pb->seq + L.roff;
				const uint8_t* pq = pb->qual + L.roff;
				BT_NOUNROLL
				for (uint32_t base = 0; base < L.plen; base += 16u) bt_rl_store_chunk(S, base, bt_ld4(ps + base), bt_ld4(pq + base));
			}
		}
	}

	for (;;) {
		/* keep the compiler from hoisting the cold descriptor's fields into scalar registers for
		 * the whole loop: they are read where they are used */
		asm volatile("" : "+s"(cold));

#ifdef BT_TRACE
		if (A.trace && L.state != ST_IDLE && L.rd == A.traceRead) {
			const uint32_t k = atomicAdd(A.trace, 1u);
			if (k < A.traceCap) {
				uint32_t* t = A.trace + 4u + 12u * k;
				t[0] = sc_rounds; t[1] = L.state; t[2] = L.step | (L.mirror << 8) | (L.readFw << 9) | (L.rev << 10) | (L.bid << 12);
				t[3] = req.kind; t[4] = req.n; t[5] = (uint32_t)req.a; t[6] = (uint32_t)(req.a >> 32);
				t[7] = (uint32_t)req.x; t[8] = (uint32_t)(req.x >> 32); t[9] = L.top; t[10] = L.bot; t[11] = L.d | (L.sd << 16);
			}
		}
#endif
		/* ---- the round's memory requests: every lane's loads are issued, then one wait ---------- */
		BT_PROF_T0(t_rank);
		BtRes res;
		{
			const bool isRank = req.kind == RQ_RANK;
			const bool isFetch = req.kind == RQ_FETCH;
			const bool m = L.mirror != 0;
			const uint8_t* blk = m ? A.H.blk[1] : A.H.blk[0];
			const uint32_t zBlk = m ? A.H.zBlk[1] : A.H.zBlk[0], zPos = m ? A.H.zPos[1] : A.H.zPos[0];
#if BT_WIDE
			const bt_row rowA = req.a, rowB = req.x;
			const uint64_t bA = rowA / BT_BLK_ROWS, bB = rowB / BT_BLK_ROWS;
#else
			const uint32_t rowA = (uint32_t)req.a, rowB = (uint32_t)req.x;
			const uint32_t bA = rowA / BT_BLK_ROWS, bB = rowB / BT_BLK_ROWS;
#endif
			const bool hasB = isRank && req.n == 2;
			/* pieces 0,1: rank row A's block, or the first two pieces of a fetch; pieces 2,3: row B's block, or the
			 * fetch's third and fourth piece */
			const uint8_t* pA = isRank ? blk + (uint64_t)bA * BT_BLK_BYTES : (const uint8_t*)(uintptr_t)req.a;
			const uint8_t* pB = hasB ? blk + (uint64_t)bB * BT_BLK_BYTES : pA + 32;
			const uint32_t nA = isRank ? 4u - (hasB ? 0u : 2u) : (isFetch ? req.n : 0u);
			const bool hasW = !RL && isRank && req.wchunk != 0xffffu;   /* next read window rides along (register-window build) */
			const bool hasX = (isFetch && req.x != 0) || hasW;
			const uint8_t* pX = hasW ? A.H.qual + L.roff + (uint64_t)req.wchunk * 16u : (const uint8_t*)(uintptr_t)req.x;
			const uint8_t* pW = A.H.seq + L.roff + (uint64_t)req.wchunk * 16u;
			BtU4 qa[4] = {}, qx = {}, qw = {};
			/* every address a lane can ask for is global memory (index, scratch arenas, ftab, SA sample,
			 * reads): plain global loads, not FLAT ones (see BT_GP) */
			/* (a text window's address is a word's, not a 16-byte piece's: bt_ld4w) */
			if (nA > 0u) qa[0] = bt_ld4w(pA);
			if (nA > 1u) qa[1] = bt_ld4w(pA + 16);
			if (nA > 2u) qa[2] = bt_ld4(pB);
			if (nA > 3u) qa[3] = bt_ld4(pB + 16);
			if (hasX) qx = bt_ld4(pX);
			if (hasW) qw = bt_ld4(pW);
#if BT_WIDE
			if (isRank) {
				/* the blocks' counters count from their segment's start: add the segment's absolute counts (a small table,
				 * cache-resident) and lay the quartets out as a fetched range-stack entry's (BtRes) */
				const uint64_t* seg = m ? A.H.segBase[1] : A.H.segBase[0];
				uint32_t lf[4], la, dummy;
				bt_rank4_blk(qa[0], qa[1], (uint32_t)(rowA % BT_BLK_ROWS), (uint32_t)bA == zBlk, zPos, lf, &la);
				{
					const uint64_t* sb = seg + (bA >> A.H.segShift) * 4u;
					const BtU4 s0 = bt_ld4(sb), s1 = bt_ld4(sb + 2);
					const uint64_t v0 = (((uint64_t)s0.y << 32) | s0.x) + lf[0], v1 = (((uint64_t)s0.w << 32) | s0.z) + lf[1];
					const uint64_t v2 = (((uint64_t)s1.y << 32) | s1.x) + lf[2], v3 = (((uint64_t)s1.w << 32) | s1.z) + lf[3];
					res.q[0].x = (uint32_t)v0; res.q[0].y = (uint32_t)(v0 >> 32); res.q[0].z = (uint32_t)v1; res.q[0].w = (uint32_t)(v1 >> 32);
					res.q[1].x = (uint32_t)v2; res.q[1].y = (uint32_t)(v2 >> 32); res.q[1].z = (uint32_t)v3; res.q[1].w = (uint32_t)(v3 >> 32);
				}
				res.x = qx; res.x.x = la;
				if (hasB) {
					bt_rank4_blk(qa[2], qa[3], (uint32_t)(rowB % BT_BLK_ROWS), (uint32_t)bB == zBlk, zPos, lf, &dummy);
					const uint64_t* sb = seg + (bB >> A.H.segShift) * 4u;
					const BtU4 s0 = bt_ld4(sb), s1 = bt_ld4(sb + 2);
					const uint64_t v0 = (((uint64_t)s0.y << 32) | s0.x) + lf[0], v1 = (((uint64_t)s0.w << 32) | s0.z) + lf[1];
					const uint64_t v2 = (((uint64_t)s1.y << 32) | s1.x) + lf[2], v3 = (((uint64_t)s1.w << 32) | s1.z) + lf[3];
					res.q[2].x = (uint32_t)v0; res.q[2].y = (uint32_t)(v0 >> 32); res.q[2].z = (uint32_t)v1; res.q[2].w = (uint32_t)(v1 >> 32);
					res.q[3].x = (uint32_t)v2; res.q[3].y = (uint32_t)(v2 >> 32); res.q[3].z = (uint32_t)v3; res.q[3].w = (uint32_t)(v3 >> 32);
				}
			} else {
				res.q[0] = qa[0]; res.q[1] = qa[1]; res.q[2] = qa[2]; res.q[3] = qa[3]; res.x = qx;
			}
#else
			if (isRank) {
				uint32_t lf[4], la;
				bt_rank4_blk(qa[0], qa[1], rowA % BT_BLK_ROWS, bA == zBlk, zPos, lf, &la);
				res.q[0].x = lf[0]; res.q[0].y = lf[1]; res.q[0].z = lf[2]; res.q[0].w = lf[3];
				res.q[2].x = la;
				res.q[3] = qw; res.x = qx;
				if (hasB) {
					uint32_t dummy;
					bt_rank4_blk(qa[2], qa[3], rowB % BT_BLK_ROWS, bB == zBlk, zPos, lf, &dummy);
					res.q[1].x = lf[0]; res.q[1].y = lf[1]; res.q[1].z = lf[2]; res.q[1].w = lf[3];
				}
			} else {
				res.q[0] = qa[0]; res.q[1] = qa[1]; res.q[2] = qa[2]; res.q[3] = qa[3]; res.x = qx;
			}
#endif
		}
		BT_PROF_ADD(PS_RANK, t_rank);

		/* ---- advance every lane to its next request, pulling new reads as old ones finish -------
		 * (A lane that finishes takes its next read at once: leaving it idle until the next round's pass was measured in the
		 * eighth and ninth GPU calls of round 5 -- 15.0-15.1 M reads/s against 15.3 M -- and dropped.) */
		req.tally = 0;
		for (;;) {
			if (L.state == ST_IDLE) {
				if (drained) break;
				const uint32_t w = atomicAdd(A.nextRead, 1u);
				if (w >= nReads) { drained = true; break; }
				BT_PROF_T0(t_refill);
				bt_lane_start<RL>(L, PROG, A.H, *cold, S, (EXT && A.order) ? BT_GP(const uint32_t, A.order)[w] : w);
				BT_PROF_ADD(PS_REFILL, t_refill);
			}
			BT_PROF_T0(t_loop);
			bt_lane_run<RL>(L, PROG, A.H, WARM, *cold, S, res, req, CNT);
			BT_PROF_ADD(PS_LOOP, t_loop);
			if (L.state == ST_IDLE) continue;
			break;
		}
		/* the wavefront leaves the loop as a whole (keeps the tallies below wave-uniform); lanes that
		 * have run out of work simply carry an empty request */
		tl_acc += req.tally;
		const bool live = L.state != ST_IDLE;
		if (!live) { req.kind = RQ_NONE; req.wchunk = 0xffffu; }
		if (__ballot(live) == 0) break;
		/* op counters: wave-uniform tallies in scalar registers (ballot + popcount), flushed once
		 * per wavefront at the end */
		{
			const bool isR = req.kind == RQ_RANK;
			sc_iters += (uint32_t)__builtin_popcountll(__ballot(live));
			sc_rounds += 1u;
			sc_fetch += (uint32_t)__builtin_popcountll(__ballot(req.kind == RQ_FETCH));
			sc_chase += (uint32_t)__builtin_popcountll(__ballot(isR && L.lfk == LFK_CHASE));
			sc_lfex += (uint32_t)__builtin_popcountll(__ballot(isR && L.lfk == LFK_EX2));
			sc_lf2 += (uint32_t)__builtin_popcountll(__ballot(isR && L.lfk == LFK_C2));
			sc_lf1 += (uint32_t)__builtin_popcountll(__ballot(isR && L.lfk == LFK_LF1));
			sc_same += (uint32_t)__builtin_popcountll(__ballot(isR && req.n == 2 && (bt_row)req.a / 448u == (bt_row)req.x / 448u));
			if (RL || WARM.locOn) {
				const bool isF = req.kind == RQ_FETCH;
				sc_locrec += (uint32_t)__builtin_popcountll(__ballot(isF && (L.state == ST_LOC_REC || (L.state == ST_RESOLVE_DONE && WARM.locOn))));
				sc_txtwin += (uint32_t)__builtin_popcountll(__ballot(isF && L.state == ST_LOC_TXT));
				/* a call adds at most a read's length to a half: flush long before 2^15 */
				if ((sc_rounds & 63u) == 63u && __ballot((tl_acc & 0x60006000u) != 0) != 0) {
					atomicAdd(&CNT[CN_TLFEX], (unsigned long long)(tl_acc >> 16)); atomicAdd(&CNT[CN_TLF1], (unsigned long long)(tl_acc & 0xffffu));
					tl_acc = 0;
				}
			}
#ifdef BT_PROFILE
			/* how much of the search runs on ranges that are one BWT row (round 5: the measurement behind locus mode) */
			{
				const bool sEx = isR && L.lfk == LFK_EX2 && (uint32_t)req.x == (uint32_t)req.a + 1u;
				const bool sAny = sEx || (isR && L.lfk == LFK_LF1);
				const unsigned long long bEx = __ballot(sEx), bRun = __ballot(sAny && !pf_prevSingle);
				if ((threadIdx.x & 63u) == 0) { atomicAdd(&CNT[CN_N + PS_SINGLE_LFEX], (unsigned long long)__builtin_popcountll(bEx)); atomicAdd(&CNT[CN_N + PS_SINGLE_RUNS], (unsigned long long)__builtin_popcountll(bRun)); }
				if (isR) pf_prevSingle = sAny;
			}
#endif
		}
		if (live) L.iters++;
		if (EXT && A.pool && A.park) {
			/* the cursor is dry (a lane of this wavefront found it so, or a look at it every 16th round says so): the
			 * wavefront stops here and parks what its lanes are doing (below, after the loop) -- unless one of them
			 * holds a read that has been carried long enough: then it runs on until that read is done */
			bool dry = __ballot(drained) != 0;
			if (!dry && (sc_rounds & 15u) == 15u) dry = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(A.nextRead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= (int)nReads;
			if (dry) {
				drained = true;
				if (sc_rounds >= A.parkMinRounds && __ballot(live && ((cold->curBid - L.bid) & (BT_BATCH_RING - 1u)) >= A.maxAge) == 0) {
					{ parkNow = true; break; }
				}
			}
		}
	}
	if (EXT && parkNow && L.state != ST_IDLE) {
		BtPoolRec* r = A.pool + (blockIdx.x * BT_BLOCK + threadIdx.x);
		uint32_t lw[4 * LANE_PIECES] = {};
		__builtin_memcpy(lw, &L, RL ? offsetof(BtLane, cs0) : sizeof(L));
		BT_UNROLL
		for (int k = 0; k < PIECES; k++) { BtU4 v; v.x = lw[4 * k]; v.y = lw[4 * k + 1]; v.z = lw[4 * k + 2]; v.w = lw[4 * k + 3]; bt_st4((uint8_t*)r->w + 16 * k, v); }
		{ BtU4 v; v.x = req.kind; v.y = req.n; v.z = req.wchunk; v.w = 0; bt_st4((uint8_t*)r->w + 16 * BT_POOL_REQ, v); }
		{ BtU4 v; v.x = (uint32_t)req.a; v.y = (uint32_t)(req.a >> 32); v.z = (uint32_t)req.x; v.w = (uint32_t)(req.x >> 32); bt_st4((uint8_t*)r->w + 16 * (BT_POOL_REQ + 1), v); }
		{ BtU4 v; v.x = A.launchSeq; v.y = 0; v.z = 0; v.w = 0; bt_st4((uint8_t*)r->w + 16 * (BT_POOL_REQ + 2), v); }
		atomicAdd(A.parkedOf + L.bid, 1u);
	}
	if (tl_acc) { atomicAdd(&CNT[CN_TLFEX], (unsigned long long)(tl_acc >> 16)); atomicAdd(&CNT[CN_TLF1], (unsigned long long)(tl_acc & 0xffffu)); }
	if ((threadIdx.x & 63u) == 0) {
		atomicAdd(&CNT[CN_LOCREC], (unsigned long long)sc_locrec); atomicAdd(&CNT[CN_TXTWIN], (unsigned long long)sc_txtwin);
		atomicAdd(&CNT[CN_ITERS], (unsigned long long)sc_iters); atomicAdd(&CNT[CN_WROUNDS], (unsigned long long)sc_rounds);
		atomicAdd(&CNT[CN_FETCH], (unsigned long long)sc_fetch); atomicAdd(&CNT[CN_CHASE], (unsigned long long)sc_chase);
		atomicAdd(&CNT[CN_LFEX], (unsigned long long)sc_lfex); atomicAdd(&CNT[CN_LF2], (unsigned long long)sc_lf2);
		atomicAdd(&CNT[CN_LF1], (unsigned long long)sc_lf1); atomicAdd(&CNT[CN_SAMEPAIR], (unsigned long long)sc_same);
	}

	__syncthreads();
	if (threadIdx.x < CN_N + PS_N && A.counts) atomicAdd(&A.counts[threadIdx.x], CNT[threadIdx.x]);
}


/* sides != 0: rank from the index files' side layout instead of the rank blocks the search uses */
__global__ void bt_probe_rank_kernel(BtIndexDev ix, const bt_row* rows, uint32_t n, bt_row* lf, uint8_t* Lout, uint32_t sides)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	bt_row r[4]; uint32_t L;
#if BT_WIDE
	bt_rank4(ix, rows[i], r, &L);
#else
	if (sides) bt_rank4_sides(ix, rows[i], r, &L); else bt_rank4(ix, rows[i], r, &L);
#endif
	lf[i * 4 + 0] = r[0]; lf[i * 4 + 1] = r[1]; lf[i * 4 + 2] = r[2]; lf[i * 4 + 3] = r[3];
	Lout[i] = (uint8_t)L;
}

__global__ void bt_probe_chase_kernel(BtIndexDev ix, const bt_row* rows, uint32_t n, uint32_t qlen,
                                      bt_row* joined, uint32_t* tidx, uint32_t* toff)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	bt_row row = rows[i]; uint32_t jumps = 0;
	while ((row & ix.offMask) != row && row != ix.zOff) {
		bt_row r[4]; uint32_t L;
		bt_rank4(ix, row, r, &L);
		row = r[L];
		jumps++;
	}
	const bt_row off = (row == ix.zOff) ? jumps : ix.offs[row >> ix.offRate] + jumps;
	joined[i] = off;
	uint32_t t = 0xffffffffu, o = 0, probes = 0;
	if (!bt_joined_to_text(ix, qlen, off, &t, &o, &probes)) { t = 0xffffffffu; o = 0; }
	tidx[i] = t; toff[i] = o;
}

/* Random-gather ceiling: every thread does `iters` independent rank queries at pseudo-random rows of
 * the index -- the memory access pattern of the search kernels with all the search logic taken away.  Each query
 * touches one 32-byte rank block (dep bit 1 clear; what the search kernels gather), or one aligned 128-byte side pair
 * of the index files' layout (dep bit 1 set: 4 x 16-byte loads of the row's side + 8 counter bytes of the partner
 * side).  dep bit 0 makes each query's row depend on the previous result (an SA walk's dependency chain) instead of
 * being known up front. */
#if BT_WIDE
extern "C" int bt_launch_gather_bench(const BtIndexDev*, uint32_t, uint32_t, uint32_t, uint32_t*, void*) { return -1; }
#else
__global__ __launch_bounds__(256) void bt_gather_bench_kernel(BtIndexDev ix, uint32_t iters, uint32_t dep, uint32_t* sink)
{
	uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
	for (uint32_t k = 0; k < iters; k++) {
		x ^= x << 13; x ^= x >> 17; x ^= x << 5;
		const uint32_t row = (uint32_t)(((uint64_t)x * ix.len) >> 32);
		uint32_t r[4], L;
		if (dep & 2u) bt_rank4_sides(ix, row, r, &L); else bt_rank4(ix, row, r, &L);
		acc += r[L];
		if (dep & 1u) x += r[L];
	}
	if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" int bt_launch_gather_bench(const BtIndexDev* ix, uint32_t nBlocks, uint32_t iters, uint32_t dep, uint32_t* sink, void* stream)
{
	hipLaunchKernelGGL(bt_gather_bench_kernel, dim3(nBlocks), dim3(256), 0, (hipStream_t)stream, *ix, iters, dep, sink);
	return (int)hipGetLastError();
}
#endif

/* ---- launchers (called from bt_api.cpp, which is plain C++) ------------------------------- */
/* occ = waves per SIMD the register allocator was told to fit (1..4): the same source compiled for
 * different register budgets; which is fastest is a measured choice (bt_api.cpp, BT_OCC). */
/* rl = every read of the batch has <= BT_RL_MAXLEN bases: the build that keeps each lane's whole read
 * in LDS (no read-window fetches); otherwise the register-window build.  rl == 2 (reads of <= BT_RL3_MAXLEN
 * bases, occ 3): the three-blocks-per-CU diet of the same. */
extern "C" int bt_launch_search(const BtKernelArgs* a, uint32_t nBlocks, int occ, int rl, void* stream)
{
	hipStream_t st = (hipStream_t)stream;
	const bool ext = a->pool || a->order || (rl & BT_RL_FORCE_EXT) != 0;   /* (diagnostics: BT_FORCE_EXT, read by the context) */
	rl &= ~BT_RL_FORCE_EXT;
#define BT_LAUNCH(O, R, T) do { if (ext) hipLaunchKernelGGL((bt_search_kernel<O, true, R, T>), dim3(nBlocks), dim3(BT_BLOCK), 0, st, *a); \
                                else hipLaunchKernelGGL((bt_search_kernel<O, false, R, T>), dim3(nBlocks), dim3(BT_BLOCK), 0, st, *a); } while (0)
	if (rl == 2) {
		BT_LAUNCH(3, true, true);
	} else if (rl) {
		/* the read copies take 42 KB of LDS per block: two blocks per CU at most */
		if (occ == 1) BT_LAUNCH(1, true, false); else BT_LAUNCH(2, true, false);
	} else {
		switch (occ) {
		case 1:  BT_LAUNCH(1, false, false); break;
		case 2:  BT_LAUNCH(2, false, false); break;
		case 3:  BT_LAUNCH(3, false, false); break;
		default: BT_LAUNCH(4, false, false); break;
		}
	}
#undef BT_LAUNCH
	return (int)hipGetLastError();
}
__global__ void bt_maxlen_kernel(const uint16_t* len, uint32_t n, uint32_t* out)
{
	uint32_t m = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t v = len[i]; m = v > m ? v : m; }
	for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, o); m = t > m ? t : m; }
	if ((threadIdx.x & 63u) == 0) atomicMax(out, m);
}
extern "C" int bt_launch_maxlen(const uint16_t* len, uint32_t n, uint32_t* out, void* stream)
{
	uint32_t nb = (n + 255u) / 256u; if (nb > 2048u) nb = 2048u;
	hipLaunchKernelGGL(bt_maxlen_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, len, n, out);
	return (int)hipGetLastError();
}
extern "C" int bt_launch_probe_rank(const BtIndexDev* ix, const bt_row* rows, uint32_t n, bt_row* lf,
                                    uint8_t* L, uint32_t sides, void* stream)
{
	hipLaunchKernelGGL(bt_probe_rank_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   *ix, rows, n, lf, L, sides);
	return (int)hipGetLastError();
}
extern "C" int bt_launch_probe_chase(const BtIndexDev* ix, const bt_row* rows, uint32_t n, uint32_t qlen,
                                     bt_row* joined, uint32_t* tidx, uint32_t* toff, void* stream)
{
	hipLaunchKernelGGL(bt_probe_chase_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   *ix, rows, n, qlen, joined, tidx, toff);
	return (int)hipGetLastError();
}
