/*
 * bt_best_kernels.hip -- gfx950 kernel of the best-first search path (--best, --strata, -M, -v 3).
 *
 *   bt_best_kernel        : the wavefront automaton of bt_best.h -- one loop per wavefront over the engine's resumable
 *                           pieces: hot rounds for the lanes that extend a branch or walk the suffix array, cold sweeps
 *                           for the rest, reads taken lane by lane from a global cursor.  Picked for indexes that do not
 *                           fit the Infinity Cache (bt_api.cpp).
 *   bt_best_nested_kernel : one lane = one read (or pair), run start to finish call by call; reads handed out a wavefront
 *                           at a time.  Small genomes, and PairedBWAlignerV1's runner always.
 *   Every lane owns an arena of `arenaWords` 32-bit words in HBM for the read's branches, heaps and driver records.  With
 *   a paired program a "read" is a pair: both mates' drivers compete in one queue and the second mate is found by scanning
 *   the 2-bit reference next to the first one's hit (PairedBWAlignerV2, aligner.h:1483-2051).  A second, small launch of
 *   either kernel (work list, 16 MB arenas, one lane per wavefront taking reads) searches again what outgrew its arena.
 *
 * Replaces (reference, CPU): the *Stateful worker loops of ebwt_search.cpp:1223/1509/1955/2609 for
 * unpaired reads (MixedMultiAligner::run + UnpairedAlignerV2, aligner.h:244-360, 381-599).
 */
#include <hip/hip_runtime.h>
#if defined(BF_PROFILE)
__device__ unsigned long long bf_prof[3 * 32];      /* bt_best.h: cycles, passes, lanes per section (BP_*; BF_PROF_SLOTS) */
#endif
#include "bt_best.h"
#include "bt_kernels.h"

/* Blocks per CU = waves per SIMD the register allocator is asked to fit.  Every lane runs its own control flow with its
 * loads where the data is needed, so what the kernel lives on is waves to switch to while one waits: measured on the
 * e_coli best-first workloads (profiles/r3/best_occupancy.txt), 1 wave per SIMD (all 512 registers, no spill) 2.87 /
 * 6.78 M reads/s single-end / paired, 3 waves 5.15 / 12.26, 4 waves 5.94 / 13.22, 6 waves (80 registers, the rest
 * spilled to scratch, which is coalesced and cached) 6.20 / 13.90, 8 waves 5.62 / 14.16.  -DBT_BEST_MIN_BLOCKS=<n>
 * builds another (make bestsweep). */
/* Default: four for the call-by-call kernel.  Six is a few per cent faster on e_coli, but every lane owns an arena (bt_api.cpp:
 * 256 KB since the hg19-scale measurement of profiles/r3/best_arena.txt), and four blocks per CU keep that at 67 GB per context.
 * The wavefront automaton (the kernel of the indexes that do not fit the caches) runs THREE blocks per CU since round 6: with
 * the leaf's state in LDS, 168 registers instead of 128 are worth more than the fourth wave (profiles/r6/call4_*: config 5's
 * share 12.15 -> 12.43 M reads/s, --best single-end 2.39 -> 2.49 M), and its arenas are 51 GB instead of 67. */
#ifndef BT_BEST_MIN_BLOCKS
#define BT_BEST_MIN_BLOCKS 4
#endif
#ifndef BT_BEST_AUTO_BLOCKS
#define BT_BEST_AUTO_BLOCKS 3
#endif
#define BT_BEST_BOUNDS __launch_bounds__(BT_BLOCK, BT_BEST_MIN_BLOCKS)
#define BT_BEST_AUTO_BOUNDS __launch_bounds__(BT_BLOCK, BT_BEST_AUTO_BLOCKS)
#ifndef BT_BEST_LEAF_LDS
/* 0: round 5's kernel, the leaf's state in scratch memory with the rest (A/B; and the build with 64-bit rows, whose leaf
 * record is 43 words: four blocks of those do not fit a CU's LDS) */
#define BT_BEST_LEAF_LDS (BT_WIDE ? 0 : 1)
#endif
/* what both kernels begin with: the batch's descriptors into LDS, the lane's record */
#define BT_BEST_PROLOGUE \
	__shared__ BfProgram PROG; \
	__shared__ BtIndexDev IX[2]; \
	__shared__ BtBatchDev BATCH; \
	__shared__ BtRefDev REF; \
	for (uint32_t i = threadIdx.x; i < sizeof(BfProgram) / 4; i += blockDim.x) ((uint32_t*)&PROG)[i] = ((const uint32_t*)A.prog)[i]; \
	for (uint32_t i = threadIdx.x; i < 2 * sizeof(BtIndexDev) / 4; i += blockDim.x) ((uint32_t*)IX)[i] = ((const uint32_t*)A.ix)[i]; \
	for (uint32_t i = threadIdx.x; i < sizeof(BtBatchDev) / 4; i += blockDim.x) ((uint32_t*)&BATCH)[i] = ((const uint32_t*)A.batch)[i]; \
	if (A.ref) for (uint32_t i = threadIdx.x; i < sizeof(BtRefDev) / 4; i += blockDim.x) ((uint32_t*)&REF)[i] = ((const uint32_t*)A.ref)[i]; \
	__syncthreads(); \
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; \
	BfLane X; \
	__builtin_memset(&X, 0, sizeof(X)); \
	const uint32_t laneStride = A.laneStride > 1u ? A.laneStride : 1u; \
	const bool laneOn = (threadIdx.x % laneStride) == 0u; \
	X.A = (BF_G uint32_t*)(A.arenas + (uint64_t)(g / laneStride) * A.arenaWords); \
	X.cap = A.arenaWords; \
	X.ix = IX; X.P = &PROG; X.ref = &REF; \
	const bool paired = PROG.paired != 0; \
	const uint32_t n = A.workList ? (*A.workCount < A.workCap ? *A.workCount : A.workCap) : BATCH.n_reads
#define BT_BEST_EPILOGUE \
	if (A.counts) { \
		/* op counters (bt_op_counts order: lfex lf2 lf1 chase ftab offs rstarts frames lane_iters same_pair) */ \
		atomicAdd(&A.counts[CN_LFEX], (unsigned long long)X.c_lfex); atomicAdd(&A.counts[CN_LF2], (unsigned long long)X.c_lf2); \
		atomicAdd(&A.counts[CN_LF1], (unsigned long long)X.c_lf1); atomicAdd(&A.counts[CN_CHASE], (unsigned long long)X.c_chase); \
		atomicAdd(&A.counts[CN_FTAB], (unsigned long long)X.c_ftab); atomicAdd(&A.counts[CN_OFFS], (unsigned long long)X.c_offs); \
		atomicAdd(&A.counts[CN_RSTARTS], (unsigned long long)X.c_rst); atomicAdd(&A.counts[CN_FRAMES], (unsigned long long)X.c_frames); \
		atomicAdd(&A.counts[CN_SAMEPAIR], (unsigned long long)X.c_same); \
	}

/* every lane its read from start to finish, reads handed out a wavefront at a time: PairedBWAlignerV1's runner, and the
 * others' for comparison (BT_BEST_NESTED=1) */
__global__ BT_BEST_BOUNDS void bt_best_nested_kernel(BtBestArgs A)
{
	BT_BEST_PROLOGUE;
	if (laneOn) for (;;) {
		const uint32_t w = atomicAdd(A.nextRead, 1u);
		if (w >= n) break;
		const uint32_t rd = A.workList ? A.workList[w] : w;
		/* a read that outgrows its arena is searched again by the host through the twin context:
		 * its partial work is not tallied */
		const BfLane before = X;
		BF_PT0(t_run);
#if BF_HAVE_V1
		if (paired && BF_IS_V1(PROG)) bf_run_pair_v1(X, BATCH, rd); else
#endif
		if (paired) bf_run_pair(X, BATCH, rd); else bf_run_read(X, BATCH, rd);
		BF_PADD(BP_RUN, t_run);
		if (X.status & BT_STF_OVERFLOW) {
			X.c_lfex = before.c_lfex; X.c_lf2 = before.c_lf2; X.c_lf1 = before.c_lf1; X.c_chase = before.c_chase;
			X.c_ftab = before.c_ftab; X.c_offs = before.c_offs; X.c_rst = before.c_rst; X.c_same = before.c_same;
			X.c_frames = before.c_frames;
		}
	}
	BT_BEST_EPILOGUE
}

/* the wavefront automaton (bt_best.h): one loop per wavefront, hot rounds for the lanes that extend a branch or walk the
 * suffix array, a cold sweep for the rest when enough of them wait for one; a lane takes its next read in the sweep */
__global__ BT_BEST_AUTO_BOUNDS void bt_best_kernel(BtBestArgs A)
{
	BT_BEST_PROLOGUE;
	BfAuto S;
	__builtin_memset(&S, 0, sizeof(S));
#if BT_BEST_LEAF_LDS
	/* the leaf's state in LDS (bt_best.h: BfAuto::leafp), 37 words per lane: with four blocks on a CU, 148 of its 160 KB */
	__shared__ uint32_t LEAF[BT_BLOCK * BF_LEAF_STRIDE];
	S.leafp = (BfLeafSt*)(LEAF + threadIdx.x * BF_LEAF_STRIDE);
	/* (gfx950 hands LDS out in 1 280-byte granules, 128 to a CU: BT_BEST_AUTO_BLOCKS blocks must fit) */
	static_assert(sizeof(LEAF) + sizeof(BfProgram) + 2 * sizeof(BtIndexDev) + sizeof(BtBatchDev) + sizeof(BtRefDev) + 64 <= (128u / BT_BEST_AUTO_BLOCKS) * 1280u,
	              "the leaf states and the descriptors must fit the block's share of the CU's LDS");
#else
	BfLeafSt leafHere;
	S.leafp = &leafHere;
#endif
	S.phase = laneOn ? BA_TAKE : BA_IDLE;
	S.kind = paired ? 2u : 1u;
	const uint32_t coldMin = A.coldMin ? A.coldMin : 1u, takeMin = A.takeMin ? A.takeMin : 1u;
	const uint32_t sendPeriod = A.sendPeriod ? A.sendPeriod : 1u, sendMin = A.sendMin ? A.sendMin : 1u;
	uint32_t round = 0;
	auto take = [&]() -> uint32_t {
		const uint32_t w = atomicAdd(A.nextRead, 1u);
		if (w >= n) return 0xffffffffu;
		return A.workList ? A.workList[w] : w;
	};
	BF_PT0(t_run);
	for (;;) {
		const bool hot = BA_IS_HOT(S.phase);
		const unsigned long long hotM = __ballot(hot), coldM = __ballot(!hot && S.phase != BA_IDLE);
		if (!hotM && !coldM) break;
		/* lanes that wait for a read take one when enough of them do, or when nothing else is left to do; a cold sweep is
		 * due when enough lanes have something to do in it (lanes that wait for a read count only when they may take one:
		 * a sweep that could do nothing for them must not keep the hot lanes from their rounds) */
		const unsigned long long takeM = __ballot(S.phase == BA_TAKE || S.phase == BA_END);
		const bool takeOk = (uint32_t)__builtin_popcountll(takeM) >= takeMin || takeM == (hotM | coldM);
		const uint32_t nSweep = (uint32_t)__builtin_popcountll(takeOk ? coldM : (coldM & ~takeM));
		if (hotM && nSweep < coldMin) {
			round++;
			const bool sendOk = (round % sendPeriod) == 0u || (uint32_t)__builtin_popcountll(__ballot(S.phase == BA_SEND)) >= sendMin;
			if (hot) (void)bf_auto_hot(X, S, sendOk);
			continue;
		}
		if (!hot) bf_auto_cold(X, BATCH, S, takeOk, take);
		/* the lanes the pass left in the middle (a driver's advance that ended without a leaf) go on in the next sweep -- or in a
		 * second pass right away (sweepTwice), which the wave model prices at more than their waiting costs */
		if (A.sweepTwice && __ballot(BA_IS_PENDING(S.phase)) != 0) { if (BA_IS_PENDING(S.phase)) bf_auto_cold(X, BATCH, S, false, take); }
	}
	BF_PADD(BP_RUN, t_run);
	BT_BEST_EPILOGUE
}

/* indices of the reads whose status carries `flag`, at most `cap` of them (*count keeps counting) */
__global__ void bt_collect_flagged_kernel(const uint8_t* status, uint32_t n, uint32_t flag, uint32_t* list, uint32_t* count, uint32_t cap)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && (status[i] & flag)) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) list[k] = i; }
}
extern "C" int bt_launch_collect_flagged(const uint8_t* status, uint32_t n, uint32_t flag, uint32_t* list, uint32_t* count, uint32_t cap, void* stream)
{
	hipLaunchKernelGGL(bt_collect_flagged_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, status, n, flag, list, count, cap);
	return (int)hipGetLastError();
}

/* the profiling build's section tallies (bt_best.h, BF_PROFILE): out[3 * n] = cycles, passes, lanes per section; reset != 0
 * clears them.  Returns the number of sections, 0 in a build without them. */
extern "C" int bt_best_prof_read(unsigned long long* out, int cap, int reset)
{
#if defined(BF_PROFILE)
	unsigned long long h[3 * BF_PROF_SLOTS] = {};
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bf_prof), sizeof(h)) != hipSuccess) return 0;
	for (int i = 0; i < 3 * BP_N && i < cap; i++) out[i] = h[i];
	if (reset) { unsigned long long z[3 * BF_PROF_SLOTS] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(bf_prof), z, sizeof(z)); }
	return BP_N;
#else
	(void)out; (void)cap; (void)reset;
	return 0;
#endif
}

/* blocks per CU the kernel's register budget allows (= waves per SIMD: 256-lane blocks, 4 SIMDs) */
extern "C" uint32_t bt_best_blocks_per_cu(int nested)
{
	return nested ? BT_BEST_MIN_BLOCKS : BT_BEST_AUTO_BLOCKS;
}

extern "C" int bt_launch_best(const BtBestArgs* a, uint32_t nBlocks, void* stream)
{
	if (a->nested) hipLaunchKernelGGL(bt_best_nested_kernel, dim3(nBlocks), dim3(BT_BLOCK), 0, (hipStream_t)stream, *a);
	else hipLaunchKernelGGL(bt_best_kernel, dim3(nBlocks), dim3(BT_BLOCK), 0, (hipStream_t)stream, *a);
	return (int)hipGetLastError();
}
