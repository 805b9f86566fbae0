/*
 * bt_best.h -- the best-first search engine, one lane = one read (gfx950 device code; also
 * compiles for the host, where tests/emu drives it without a GPU).
 *
 * What it computes is what the reference's stateful workers compute for a single-end read
 * (`--best`, `--strata`, `-M`, `-v 3`; ebwt_search.cpp:1223, 1509, 1955, 2609): the tree of
 * RangeSourceDrivers built by Unpaired{Exact,1mm,23mm,Seed}AlignerFactory::create()
 * (aligner_0mm.h:69, aligner_1mm.h:73, aligner_23mm.h:73, aligner_seed_mm.h:82), advanced by
 * UnpairedAlignerV2::advance (aligner.h:503-567), each leaf an EbwtRangeSource
 * (ebwt_search_backtrack.h:1788-2599) doing best-first branch-and-bound over a PathManager's
 * priority queue (range_source.h:517-1574), ranges resolved row by row by RangeChaser
 * (range_chaser.h:52-209).  Results are bit-identical, which pins down: the order of every
 * per-read LCG draw, libstdc++'s heap sift order (keys change while a Branch is queued), and the
 * Branch ids the reference derives from its pool allocator (pool.h:216-322).
 *
 * How it is laid out here (nothing like the reference's heap-allocated object graph):
 *
 *   * Every read owns one bump-allocated arena of 32-bit words in HBM (slot = lane).  All
 *     "objects" are word offsets into it; nothing is ever freed (a read's arena is recycled whole
 *     when the lane picks up its next read).
 *   * A driver is a fixed 32-word record (BF_DRW), whatever its kind (leaf / cost-aware list /
 *     seeded pair); the static part of the tree is instantiated from a small table the host
 *     compiles from the policy (BfProgram), partial-alignment extenders are appended on demand.
 *   * A Branch is a 16-word record.  Its per-position RangeStates are not stored: only positions
 *     that still have an untried substitution get a 10-word "alternative" record, appended
 *     behind the branch while it grows (exactly one branch grows at a time per read: a
 *     PathManager only yields control with a freshly split front branch), so a curtailed branch
 *     is its record plus a dense run of alternatives.  The reference keeps qlen-rdepth
 *     RangeStates per branch (range_source.h:567-576).
 *   * Edit lists are parent-linked (a branch stores its own edit and its parent's offset).
 *   * Each PathManager's heap is an array of branch offsets, doubled on demand.
 *
 * Every lane runs its own control flow (loads where the data is needed): the kernel is plain SIMT, not
 * organised into lock-step rounds the way bt_core.h's engine is.  DESIGN.md 4.2 has the measurements
 * of what that costs and of what has been done about it.
 */
#ifndef BT_BEST_H_
#define BT_BEST_H_

#include "bt_rank.h"
#include "bt_core.h"

#if defined(__HIPCC__)
#define BF_FN static __host__ __device__
#define BF_INL __host__ __device__ __forceinline__
/* The automaton's own layer as real calls (the default) or inlined into its kernel's loop (-DBF_FNI_INLINE).  As real
 * functions its pieces take the lane's records by reference, and a record whose address is handed to a call lives in scratch
 * memory for the whole kernel; inlined, the loop is one 100 k-instruction body.  Measured both ways (profiles/r5/call1_SUMMARY.txt,
 * profiles/r4/eleventh_call_inlined_layer_ecoli.txt): at hg19 scale, where the library picks the automaton, calls win
 * (BASELINE config 5's share 10.09 against 9.57 M reads/s); on e_coli with the automaton forced, inlining does (29.1 against
 * 23.7 M) -- but there the library picks the call-by-call kernel anyway. */
#ifdef BF_FNI_INLINE
#define BF_FNI static __host__ __device__ __attribute__((always_inline))
#else
#define BF_FNI static __host__ __device__
#endif
#else
#define BF_FN static
#define BF_INL static inline
#define BF_FNI static
#endif
/* PairedBWAlignerV1 (bf_run_pair_v1, the reference's default paired-end aligner): part of every build since round 3
 * (GPU-verified against the 120 reference outputs of tests/golden/pe_v1) */
#if defined(__HIP_DEVICE_COMPILE__)
#define BF_UNROLL16 _Pragma("unroll 16")
#else
#define BF_UNROLL16
#endif
#define BF_HAVE_V1 1
/* How the engine is organised around what a lane waits for (round 4: measured on the GPU against the reference-shaped
 * code it replaced, profiles/r4/first_call_SUMMARY.txt and third_call_SUMMARY.txt -- 1.9x / 1.7x on e_coli, 1.6x / 1.4x at
 * hg19 scale; the other halves of those forks are gone):
 *   * leaf_advance_branch keeps a branch it is simply extending in registers from step to step (the streak, see there);
 *   * a leaf's six PathManager words live in the lane while the leaf is worked on (pm_enter / pm_leave);
 *   * the cost-aware driver's sort and mate check work on gathered copies of their children's flags and costs;
 *   * the aligner's driver is advanced by bf_advance_top, which walks down to the leaf that is due (the first halves of
 *     cost_advance / seeded_advance), advances it at ONE place in the code, and walks back up (their second halves) -- so
 *     that the lanes of a wavefront, each with its own leaf, run their extension loops together whichever kind of node the
 *     leaf hangs from.
 * Reads are handed out a wavefront at a time (bt_best_kernels.hip): a loop that gave a lane its next read while the others
 * were still on theirs was measured 3.4x slower on e_coli (every turn then pays for some lane's tree set-up). */
#define BF_IS_V1(P) ((P).paired == 2u)
/* Section timers of the profiling build (-DBF_PROFILE, `make bestprof`; scripts/best_sections.py): wavefront cycles, passes
 * and lanes per section, tallied by the first active lane into bf_prof[] (bt_best_kernels.hip).  Sections nest (a leaf's
 * advance contains the streak, the curtail and the split); a section entered by some of a wavefront's lanes while the
 * others wait counts what the wavefront spends on those lanes.  No-ops in the product build. */
enum { BP_RUN = 0, BP_BEGIN, BP_SETQ, BP_ADV, BP_LEAF, BP_STREAK, BP_CURTAIL, BP_SPLIT, BP_SORT, BP_CHASE, BP_REPORT, BP_REF, BP_END, BP_FRONT,
       /* the wavefront automaton's own: a hot round and its pieces, a cold sweep and its pieces */
       BP_HOT, BP_HSTEP, BP_HSEND, BP_HCHASE, BP_COLD, BP_CTAKE, BP_CEXIT, BP_CPOST, BP_CRUN, BP_CPRE, BP_N };   /* BP_N <= BF_PROF_SLOTS */
#define BF_PROF_SLOTS 32
#if defined(BF_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long bf_prof[3 * BF_PROF_SLOTS];
#define BF_PT0(v) const unsigned long long v = __builtin_readcyclecounter()
#define BF_PADD(k, v) do { const unsigned long long ex_ = __ballot(1); \
	if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex_)) { atomicAdd(&bf_prof[3 * (k)], __builtin_readcyclecounter() - (v)); \
		atomicAdd(&bf_prof[3 * (k) + 1], 1ull); atomicAdd(&bf_prof[3 * (k) + 2], (unsigned long long)__builtin_popcountll(ex_)); } } while (0)
#else
#define BF_PT0(v)
#define BF_PADD(k, v)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define BF_G __attribute__((address_space(1)))
#else
#define BF_G
#endif

/* ---- the policy, compiled on the host (bt_host_compile_best) ------------------------------- */
enum { BF_PIN_BEGIN = 1, BF_PIN_LEN = 2, BF_PIN_HI_HALF = 3, BF_PIN_SEED = 4 };   /* SearchConstraintExtent */
enum { BF_LEAF = 0, BF_COST = 1, BF_SEEDED = 2 };

struct BfSpec {                   /* EbwtRangeSource + EbwtRangeSourceDriver constructor arguments */
	uint8_t  mirror;              /* 1: searches the mirror index (ebwtBw)                        */
	uint8_t  fw;                  /* read orientation                                             */
	uint8_t  reportExacts, halfAndHalf, partial;
	uint8_t  seed;                /* EbwtRangeSourceDriver::seed_: truncate the query to the seed */
	uint8_t  nudgeLeft, useBtCnt;
	uint8_t  mate;                /* 0: mate 1 / the unpaired read, 1: mate 2                     */
	uint8_t  rev[4];              /* rev0Off..rev3Off as BF_PIN_*                                 */
	uint32_t qualLim, seedLen;
};
struct BfNode { uint8_t kind, spec, genSpec, fw; };   /* child of the top-level cost-aware driver */
#define BF_MAX_SPECS 48
#define BF_MAX_NODES 16
struct BfProgram {
	BfSpec   specs[BF_MAX_SPECS];
	BfNode   nodes[BF_MAX_NODES];
	uint32_t nspecs, nnodes;
	uint32_t maq, maxBts, btCntOn, strandFix;
	uint32_t sinkN, sinkMax, sinkAll, sinkStrata, sampleMax;
	uint32_t needMirror;
	/* paired-end (PairedBWAlignerV2, aligner.h:1483-2051) */
	uint32_t paired, minIns, maxIns, mate1Fw, mate2Fw, pairTries, allowContain;
	/* the RefAligner the pair's second mate is found with (ref_aligner.h): k mismatches end to end
	 * (-v k) or k in the seed with a quality ceiling (-n k) */
	uint32_t refSeeded, refMms, refSeedLen, refQualMax;
};
/* BfProgram::paired: 0 unpaired, 1 PairedBWAlignerV2 (--best), 2 PairedBWAlignerV1 (no --best: the same nodes, grouped
 * by (mate, strand) under four cost-aware drivers; its symCeiling is -m = sinkMax / 2) */

/* the 2-bit reference with its N mask (BitPairReference, reference.h:35-120), one position space
 * for all sequences: reference t occupies positions [start[t], start[t] + len) */
struct BtRefDev {
	const uint32_t* bits;        /* 16 bases per word, base i at bits 2(i%16)                    */
	const uint32_t* nmask;       /* 32 positions per word, 1 = N / gap                           */
	const uint64_t* start;       /* [nRefs]                                                      */
	const uint32_t* approxLen;   /* [nRefs] up to the end of the last unambiguous stretch        */
	uint32_t nRefs, pad;
};

/* ---- record layouts (word indices) ----------------------------------------------------------
 * A BWT row takes BF_RW 32-bit words of a record: one, or two (low word first) in the build with 64-bit rows (bt_rank.h,
 * "the row type"); everything else is the same in both. */
#define BF_RW ((uint32_t)sizeof(bt_row) / 4u)
#define BF_DRW 32u
enum {
	DR_KIND = 0,       /* kind | fw<<8 | mate<<9 | spec<<16                                        */
	DR_FLAGS,          /* 1 done, 2 foundRange                                                     */
	DR_COST,           /* minCost | minCostAdjustment<<16                                          */
	/* leaf (EbwtRangeSourceDriver + its PathManager + its EbwtRangeSource) */
	LF_HEAP, LF_HEAPSZ /* sz | cap<<16 */, LF_BP /* bpCur | bpPool<<16 */, LF_BPLAST /* lastCur[0] | lastCur[1]<<16 */,
	LF_PMCOST, LF_RND, LF_QLEN /* qlen | len<<16 */, LF_REV01, LF_REV23, LF_D53 /* depth5 | depth3<<16 */,
	LF_RSFLAGS         /* 1 rs.done, 2 rs.foundRange, 4 skippingThisRead, 8 seedRange valid        */,
	LF_CURTOP, LF_CURBOT = LF_CURTOP + BF_RW, LF_CURCOST = LF_CURBOT + BF_RW /* cost | numMms<<16 */, LF_CURBR /* branch whose edits the range carries */,
	LF_SEED            /* seedRange: cost | n<<16 */, LF_SEEDMM0, LF_SEEDMM1, LF_SEEDMM2 /* mms | refc<<16 */,
	/* cost-aware (CostAwareRangeSourceDriver) */
	CA_RSS = 3, CA_NRSS /* n | cap<<16 */, CA_ACT, CA_NACT, CA_RND, CA_LAST, CA_DELAYED, CA_OPTS /* 1 strandFix, 2 patsrc set */,
	CA_KEY /* the sort's key array (drivers with more than sixteen children) */, CA_KEYCAP,
	/* seeded (EbwtSeededRangeSourceDriver) */
	SD_FULL = 3, SD_SEED, SD_FACT
};
#define BF_F_DONE 1u
#define BF_F_FOUND 2u

#if BT_WIDE
#define BF_BRW 20u     /* five 16-byte pieces: [id d01 d23 rdlen] [costham top] [bot flags] ... as the indices below fall */
#else
#define BF_BRW 16u
#endif
enum { BR_ID = 0, BR_D01, BR_D23, BR_RDLEN /* rdepth | len<<16 */, BR_COSTHAM /* cost | ham<<16 */, BR_TOP, BR_BOT = BR_TOP + BF_RW,
       BR_FLAGS = BR_BOT + BF_RW /* 1 curtailed 2 exhausted 4 delayedIncrease 8 ltop valid 16 lbot valid | delayedCost<<16 */,
       BR_LTOP, BR_LBOT = BR_LTOP + BF_RW, BR_ALT = BR_LBOT + BF_RW, BR_NALT, BR_PARENT, BR_EDIT /* pos | refc<<10 | nedits<<16 */, BR_HILO /* hi | lo<<16 */ };
#define BRF_CURTAILED 1u
#define BRF_EXHAUSTED 2u
#define BRF_DELAYED 4u
#define BRF_LTOP 8u
#define BRF_LBOT 16u

#define BF_ALW (8u * BF_RW + 2u)     /* tops[4] bots[4] info pad; info = i | quallo<<16 | elim mask<<24 | eliminated<<28 */
#define BF_ALI (8u * BF_RW)          /* the info word's index in an alternative's record */
#define BF_RESERVED 64u
/* AllocOnlyPool<Branch>: 256 KB chunk / sizeof(Branch) (pool.h:32,198): 136 bytes in the 32-bit build, 160 in the
 * 64-bit one (BtIndexDev::wide) */
#define BF_BPOOL_LIM(X) ((X).ix[0].wide ? 1638u : 1927u)
#define BF_ADV_COST_CHANGES 2

struct BfChase {                  /* RangeChaser + RowChaser state */
	uint32_t mirror, qlen;
	bt_row   top, bot, irow, row;
	uint32_t tidx, toff, done, cDone;
	bt_row   cRow;
	uint32_t cJumps;
	bt_row   cOff;                /* offset in the joined text */
};
#define BF_NONE32 0xffffffffu     /* "no sequence" in a chaser's tidx / toff (BT_OFF_MASK where rows are 32 bits) */

struct BfRead { BF_G const uint8_t* seq; BF_G const uint8_t* qual; uint32_t len, seed; };

struct BfLane {                   /* per-lane registers / private memory */
	BF_G uint32_t* A;             /* this read's arena                                            */
	uint32_t cap, top, ovf;
	const BtIndexDev* ix;         /* [2]: text index, mirror index                                */
	const BfProgram* P;
	const BtRefDev* ref;          /* paired-end only                                              */
	BfRead R[2];                  /* the read / the two mates                                     */
	uint32_t hasN;                /* bit m: mate m holds an N (found once, at bf_read_begin)      */
	uint32_t pm[7];               /* the leaf's PathManager words LF_HEAP..LF_RND while leaf_set_query / leaf_advance run (pm_enter / pm_leave); [6] = the queue's front */
	uint32_t rd;
	int32_t  btCnt;
	uint32_t alRnd;
	uint32_t nhits, stored, bestStratum, status;
	uint32_t growing;             /* branch whose alternatives are being appended (contiguity check) */
	uint32_t c_lfex, c_lf2, c_lf1, c_chase, c_ftab, c_offs, c_rst, c_same, c_frames;
};

#define AW(o) (X.A[(o)])
/* a row in the arena (AR) / in a register copy of a record (RR) */
#if BT_WIDE
#define AR(o) ((bt_row)(uint32_t)AW(o) | ((bt_row)(uint32_t)AW((o) + 1u) << 32))
#define AR_SET(o, v) do { const bt_row ar_v_ = (v); const uint32_t ar_o_ = (o); AW(ar_o_) = (uint32_t)ar_v_; AW(ar_o_ + 1u) = (uint32_t)(ar_v_ >> 32); } while (0)
#define RR(R, i) ((bt_row)(R)[(i)] | ((bt_row)(R)[(i) + 1u] << 32))
#define RR_SET(R, i, v) do { const bt_row rr_v_ = (v); (R)[(i)] = (uint32_t)rr_v_; (R)[(i) + 1u] = (uint32_t)(rr_v_ >> 32); } while (0)
#else
#define AR(o) AW(o)
#define AR_SET(o, v) (AW(o) = (v))
#define RR(R, i) ((R)[(i)])
#define RR_SET(R, i, v) ((R)[(i)] = (v))
#endif

BF_INL uint32_t bf_rnd(uint32_t& last)                       /* RandomSource::nextU32, random_source.h:45-54 */
{
	last = 1664525u * last + 1013904223u;
	uint32_t ret = last >> 16;
	last = 1664525u * last + 1013904223u;
	return ret ^ last;
}
BF_INL uint32_t bf_rnd_at(BfLane& X, uint32_t off) { uint32_t s = AW(off); uint32_t r = bf_rnd(s); AW(off) = s; return r; }
/* The six PathManager words of a leaf (queue base, size | capacity, branch-pool cursors, queue cost, generator) are read
 * at the head of nearly every dependent chain of loads in the leaf's code: they live in the lane for
 * the duration of leaf_set_query / leaf_advance (pm_enter loads them, pm_leave stores them; nothing else runs between). */
#define PMW(w) (X.pm[(w) - LF_HEAP])
#define PM_RND(X, d) bf_rnd(X.pm[LF_RND - LF_HEAP])
BF_INL void pm_enter(BfLane& X, uint32_t d)
{
	for (uint32_t k = 0; k < 6u; k++) X.pm[k] = AW(d + LF_HEAP + k);
	X.pm[6] = (X.pm[LF_HEAPSZ - LF_HEAP] & 0xffffu) ? (uint32_t)AW(X.pm[0]) : 0u;          /* kept equal to heap[0] by every store to it (PM_HSET) */
}
#define PM_HSET(heap, idx, x) do { const uint32_t hs_i = (idx), hs_x = (x); AW((heap) + hs_i) = hs_x; if (hs_i == 0) X.pm[6] = hs_x; } while (0)
BF_INL void pm_leave(BfLane& X, uint32_t d) { for (uint32_t k = 0; k < 6u; k++) AW(d + LF_HEAP + k) = X.pm[k]; }

BF_INL uint32_t bf_alloc(BfLane& X, uint32_t n)
{
	if (X.top + n > X.cap) { X.ovf = 1; return 0; }           /* offsets [0, BF_RESERVED) are a harmless dump */
	uint32_t o = X.top; X.top += n; return o;
}

/* Read::patFw / patRc / patFwRev / patRcRev and qual / qualRev (read.h:119-133) from the one stored copy */
BF_INL uint32_t bf_base(const BfRead& R, uint32_t fw, uint32_t ebwtFw, uint32_t i)
{
	uint32_t c = R.seq[(fw == ebwtFw) ? i : R.len - 1u - i];
	return (!fw && c < 4u) ? (c ^ 3u) : c;
}
BF_INL uint32_t bf_qualc(const BfRead& R, uint32_t fw, uint32_t ebwtFw, uint32_t i)
{
	return R.qual[(fw == ebwtFw) ? i : R.len - 1u - i];
}
BF_INL uint32_t bf_phred(uint32_t c) { return c >= 33u ? c - 33u : 0u; }

/* ---- Branch ---------------------------------------------------------------------------------- */
BF_INL uint32_t br_cost(BfLane& X, uint32_t b) { return AW(b + BR_COSTHAM) & 0xffffu; }
BF_INL uint32_t br_ham(BfLane& X, uint32_t b) { return AW(b + BR_COSTHAM) >> 16; }
BF_INL void br_set_cost(BfLane& X, uint32_t b, uint32_t c) { AW(b + BR_COSTHAM) = (AW(b + BR_COSTHAM) & 0xffff0000u) | (c & 0xffffu); }
BF_INL uint32_t br_rdepth(BfLane& X, uint32_t b) { return AW(b + BR_RDLEN) & 0xffffu; }
BF_INL uint32_t br_len(BfLane& X, uint32_t b) { return AW(b + BR_RDLEN) >> 16; }
BF_INL uint32_t br_nedits(BfLane& X, uint32_t b) { return AW(b + BR_EDIT) >> 16; }

/* Branch::prep and the locus part of Branch::init (range_source.h:559-566, 946-954) */
BF_INL void br_prep(BfLane& X, uint32_t b)
{
	const bt_row top = AR(b + BR_TOP), bot = AR(b + BR_BOT);
	uint32_t f = AW(b + BR_FLAGS);
	if (bot > top + 1u) { f |= BRF_LTOP | BRF_LBOT; AR_SET(b + BR_LTOP, top); AR_SET(b + BR_LBOT, bot); }
	else if (bot > top) { f = (f | BRF_LTOP) & ~BRF_LBOT; AR_SET(b + BR_LTOP, top); }
	AW(b + BR_FLAGS) = f;
}

/* Branch::init (range_source.h:527-604) */
BF_FN uint32_t br_new(BfLane& X, uint32_t id, uint32_t d01, uint32_t d23, uint32_t rdepth, uint32_t len, uint32_t cost,
                      uint32_t ham, bt_row top, bt_row bot, uint32_t parent, uint32_t edit, uint32_t hilo)
{
	/* branch records start on a 16-byte boundary: leaf_advance_branch reads one in four 16-byte pieces */
	if (X.top & 3u) (void)bf_alloc(X, 4u - (X.top & 3u));
	const uint32_t b = bf_alloc(X, BF_BRW);
	/* the record in four 16-byte stores, prepped from the values at hand (br_prep reads back what was just stored) */
#if BT_WIDE
	{
		/* the wide build's record (five pieces; rows are two words): word by word, prepped */
		for (uint32_t k = 0; k < BF_BRW; k++) AW(b + k) = 0;
		AW(b + BR_ID) = id; AW(b + BR_D01) = d01; AW(b + BR_D23) = d23;
		AW(b + BR_RDLEN) = rdepth | (len << 16); AW(b + BR_COSTHAM) = (cost & 0xffffu) | (ham << 16);
		AR_SET(b + BR_TOP, top); AR_SET(b + BR_BOT, bot);
		AW(b + BR_PARENT) = parent; AW(b + BR_EDIT) = edit; AW(b + BR_HILO) = hilo;
		br_prep(X, b);
	}
#else
	{
		uint32_t f = 0, lt = 0, lb = 0;
		if (bot > top + 1u) { f = BRF_LTOP | BRF_LBOT; lt = top; lb = bot; }
		else if (bot > top) { f = BRF_LTOP; lt = top; }
		BtU4 q0, q1, q2, q3;
		q0.x = id; q0.y = d01; q0.z = d23; q0.w = rdepth | (len << 16);
		q1.x = (cost & 0xffffu) | (ham << 16); q1.y = top; q1.z = bot; q1.w = f;
		q2.x = lt; q2.y = lb; q2.z = 0; q2.w = 0;
		q3.x = parent; q3.y = edit; q3.z = hilo; q3.w = 0;
		bt_st4((void*)(X.A + b), q0); bt_st4((void*)(X.A + b + 4u), q1); bt_st4((void*)(X.A + b + 8u), q2); bt_st4((void*)(X.A + b + 12u), q3);
	}
#endif
	X.c_frames++;
	return b;
}

/* lowest marginal cost over the untried alternatives of a branch: the scan Branch::curtail
 * (range_source.h:885-902) and Branch::splitBranch (:669-707) both make.  A position below depth0
 * never gets an alternative record, so the records are exactly the loop's candidates. */
BF_INL uint32_t alt_cost(uint32_t info, uint32_t rdepth, uint32_t seedLen)
{
	const uint32_t i = info & 0xffffu;
	return ((rdepth + i < seedLen) ? (1u << 14) : 0u) | ((info >> 16) & 0xffu);
}

/* ---- CostCompare (range_source.h:1103-1142) and the heap ----------------------------------------
 * std::priority_queue<Branch*, vector, CostCompare> = libstdc++'s __push_heap / __adjust_heap
 * (bits/stl_heap.h); reproduced step for step because keys change while a branch is queued. */
/* what CostCompare looks at, fetched in one go (the record's first two 16-byte pieces) instead of field by field as the
 * comparison proceeds */
struct BfKey { uint32_t cost, un, depth, id; };
BF_INL BfKey bf_key(BfLane& X, uint32_t b)
{
	const BtU4 q0 = bt_ld4((const void*)(X.A + b)), q1 = bt_ld4((const void*)(X.A + b + 4u));
	BfKey k;
	k.id = q0.x; k.depth = ((q0.w & 0xffffu) + (q0.w >> 16)) & 0xffffu;
#if BT_WIDE
	k.cost = q1.x & 0xffffu; k.un = ((uint32_t)AW(b + BR_FLAGS) & (BRF_CURTAILED | BRF_EXHAUSTED)) != 0 ? 1u : 0u;
#else
	k.cost = q1.x & 0xffffu; k.un = (q1.w & (BRF_CURTAILED | BRF_EXHAUSTED)) != 0 ? 1u : 0u;
#endif
	return k;
}
BF_INL bool bf_before_k(const BfKey& a, const BfKey& b)        /* bf_before on fetched keys */
{
	if (a.cost != b.cost) return b.cost < a.cost;
	if (b.un && !a.un) return false;
	if (a.un && !b.un) return true;
	if (a.depth != b.depth) return a.depth < b.depth;
	return b.id < a.id;
}
BF_FN void pm_push(BfLane& X, uint32_t d, uint32_t v)
{
	uint32_t heap = PMW(LF_HEAP), sz = PMW(LF_HEAPSZ) & 0xffffu, cap = PMW(LF_HEAPSZ) >> 16;
	if (sz == cap) {
		const uint32_t ncap = cap ? cap * 2u : 8u;
		if (ncap > 0xffffu) { X.ovf = 1; return; }
		const uint32_t nh = bf_alloc(X, ncap);
		if (X.ovf) return;
		for (uint32_t k = 0; k < sz; k++) AW(nh + k) = AW(heap + k);
		heap = nh; cap = ncap; PMW(LF_HEAP) = heap;
	}
	const BfKey kv = bf_key(X, v);
	uint32_t hole = sz;
	while (hole > 0) {
		const uint32_t parent = (hole - 1u) / 2u, pv = AW(heap + parent);
		if (!bf_before_k(bf_key(X, pv), kv)) break;
		PM_HSET(heap, hole, pv); hole = parent;
	}
	PM_HSET(heap, hole, v);
	PMW(LF_HEAPSZ) = (sz + 1u) | (cap << 16);
	PMW(LF_PMCOST) = hole == 0 ? kv.cost : br_cost(X, AW(heap));
}
BF_FN uint32_t pm_pop(BfLane& X, uint32_t d)
{
	const uint32_t heap = PMW(LF_HEAP), n = PMW(LF_HEAPSZ) & 0xffffu, cap = PMW(LF_HEAPSZ) >> 16;
	const uint32_t top = AW(heap);
	uint32_t frontCost = 0; bool haveFront = false;
	if (n > 1u) {
		const uint32_t value = AW(heap + n - 1u);
		PM_HSET(heap, n - 1u, top);
		const BfKey kval = bf_key(X, value);
		const uint32_t len = n - 1u;
		uint32_t hole = 0, second = 0;
		while (second < (len - 1u) / 2u) {
			second = 2u * (second + 1u);
			const uint32_t hs = AW(heap + second), hs1 = AW(heap + second - 1u);
			uint32_t pick = hs;
			if (bf_before_k(bf_key(X, hs), bf_key(X, hs1))) { second--; pick = hs1; }
			PM_HSET(heap, hole, pick); hole = second;
		}
		if ((len & 1u) == 0 && second == (len - 2u) / 2u) {
			second = 2u * (second + 1u);
			PM_HSET(heap, hole, (uint32_t)AW(heap + second - 1u)); hole = second - 1u;
		}
		while (hole > 0) {
			const uint32_t parent = (hole - 1u) / 2u, pv = AW(heap + parent);
			if (!bf_before_k(bf_key(X, pv), kval)) break;
			PM_HSET(heap, hole, pv); hole = parent;
		}
		PM_HSET(heap, hole, value);
		if (hole == 0) { frontCost = kval.cost; haveFront = true; }
	}
	PMW(LF_HEAPSZ) = (n - 1u) | (cap << 16);
	PMW(LF_PMCOST) = haveFront ? frontCost : br_cost(X, n > 1u ? AW(heap) : top);
	return top;
}

BF_INL uint32_t pm_size(BfLane& X, uint32_t d) { return PMW(LF_HEAPSZ) & 0xffffu; }
BF_INL uint32_t pm_front(BfLane& X, uint32_t d) { return X.pm[6]; }
BF_INL void pm_reset(BfLane& X, uint32_t d)                    /* PathManager::reset (range_source.h:1386-1399) */
{
	PMW(LF_HEAPSZ) &= 0xffff0000u; PMW(LF_BP) = 0; PMW(LF_BPLAST) = 0; PMW(LF_PMCOST) = 0;
}

/* AllocOnlyPool<Branch>::alloc + lastId (pool.h:216-223, 320-322, 335-352) */
BF_FN uint32_t pm_alloc_id(BfLane& X, uint32_t d)
{
	uint32_t cur = PMW(LF_BP) & 0xffffu, pool = PMW(LF_BP) >> 16;
	if (cur + 1u >= BF_BPOOL_LIM(X)) {
		if (pool >= 2u) { X.ovf = 1; return 0; }
		uint32_t last = PMW(LF_BPLAST);
		last = pool == 0 ? ((last & 0xffff0000u) | cur) : ((last & 0xffffu) | (cur << 16));
		PMW(LF_BPLAST) = last;
		pool++; cur = 0;
	}
	cur++;
	PMW(LF_BP) = cur | (pool << 16);
	return (pool << 16) | cur;
}
/* AllocOnlyPool<Branch>::free(T*) (pool.h:279-292): only the topmost element's id is given back */
BF_FN void pm_free_id(BfLane& X, uint32_t d, uint32_t id)
{
	uint32_t cur = PMW(LF_BP) & 0xffffu, pool = PMW(LF_BP) >> 16;
	if (cur > 0 && id == ((pool << 16) | cur)) {
		cur--;
		if (cur == 0 && pool > 0) { pool--; const uint32_t last = PMW(LF_BPLAST); cur = pool == 0 ? (last & 0xffffu) : (last >> 16); }
		PMW(LF_BP) = cur | (pool << 16);
	}
}

/* PathManager::curtail (range_source.h:1435-1454) */
/* PathManager::curtail (range_source.h:1402-1424) + Branch::curtail (:877-939) for the branch whose record la_send holds in
 * registers (R == what the arena has): only the alternatives' info words are fetched */
BF_FNI void pm_curtail_regs(BfLane& X, uint32_t d, uint32_t br, uint32_t seedLen, const uint32_t* R)
{
	const uint32_t alt = R[BR_ALT], n = R[BR_NALT], rdepth = R[BR_RDLEN] & 0xffffu;
	uint32_t lowest = 0xffffu;
	for (uint32_t k0 = 0; k0 < n; k0 += 8u) {
		/* eight alternatives' info words in flight at a time (they sit ten words apart) */
		uint32_t in8[8];
		for (uint32_t j = 0; j < 8u; j++) in8[j] = k0 + j < n ? (uint32_t)AW(alt + (k0 + j) * BF_ALW + BF_ALI) : (1u << 28);
		for (uint32_t j = 0; j < 8u; j++) {
			const uint32_t info = in8[j];
			if (info >> 28) continue;
			const uint32_t c = alt_cost(info, rdepth, seedLen);
			if (c < lowest) lowest = c;
		}
	}
	uint32_t f = R[BR_FLAGS];
	const uint32_t orig = R[BR_COSTHAM] & 0xffffu;
	uint32_t ncost = orig;
	if (lowest > 0 && lowest != 0xffffu) { ncost = (orig + lowest) & 0xffffu; AW(br + BR_COSTHAM) = (R[BR_COSTHAM] & 0xffff0000u) | ncost; }
	else if (lowest == 0xffffu) f |= BRF_EXHAUSTED;
	f |= BRF_CURTAILED;
	AW(br + BR_FLAGS) = f;
	if (X.growing == br) X.growing = 0;
	if (f & BRF_EXHAUSTED) { pm_pop(X, d); pm_free_id(X, d, R[BR_ID]); }
	else if (ncost != orig) { const uint32_t p = pm_pop(X, d); pm_push(X, d, p); }
}

/* br_split with the branch's record and the chosen alternative's ranges fetched in one go each */
BF_FN uint32_t br_split(BfLane& X, uint32_t d, uint32_t b, uint32_t seedLen, uint32_t depth5)
{
	const uint32_t id = pm_alloc_id(X, d);
	uint32_t P[BF_BRW];
	{
		const BtU4 r0 = bt_ld4((const void*)(X.A + b)), r1 = bt_ld4((const void*)(X.A + b + 4u));
		const BtU4 r2 = bt_ld4((const void*)(X.A + b + 8u)), r3 = bt_ld4((const void*)(X.A + b + 12u));
		P[0] = r0.x; P[1] = r0.y; P[2] = r0.z; P[3] = r0.w; P[4] = r1.x; P[5] = r1.y; P[6] = r1.z; P[7] = r1.w;
		P[8] = r2.x; P[9] = r2.y; P[10] = r2.z; P[11] = r2.w; P[12] = r3.x; P[13] = r3.y; P[14] = r3.z; P[15] = r3.w;
#if BT_WIDE
		const BtU4 r4 = bt_ld4((const void*)(X.A + b + 16u));
		P[16] = r4.x; P[17] = r4.y; P[18] = r4.z; P[19] = r4.w;
#endif
	}
	const uint32_t alt = P[BR_ALT], n = P[BR_NALT], rdepth = P[BR_RDLEN] & 0xffffu;
	uint32_t tied[3] = {0, 0, 0}, numTied = 0, numNotElim = 0, best = 0xffffu, next = 0xffffu;
	for (uint32_t k0 = 0; k0 < n; k0 += 8u) {
		uint32_t in8[8];
		for (uint32_t j = 0; j < 8u; j++) in8[j] = k0 + j < n ? (uint32_t)AW(alt + (k0 + j) * BF_ALW + BF_ALI) : (1u << 28);
		for (uint32_t j = 0; j < 8u; j++) {
			const uint32_t info = in8[j], k = k0 + j;
			if (info >> 28) continue;
			numNotElim++;
			const uint32_t c = alt_cost(info, rdepth, seedLen);
			if (c < best) { next = best; best = c; numTied = 1; tied[0] = k; }
			else if (c == best) {
				if (numTied < 3u) tied[numTied++] = k;
				else { tied[0] = tied[1]; tied[1] = tied[2]; tied[2] = k; }
			} else if (c < next) next = c;
		}
	}
	uint32_t r = 0;
	if (numTied > 1u) r = PM_RND(X, d) % numTied;
	const uint32_t rec = alt + tied[r] * BF_ALW;
#if BT_WIDE
	bt_row W[8];
	for (uint32_t k = 0; k < 8u; k++) W[k] = AR(rec + k * BF_RW);
	uint32_t info = AW(rec + BF_ALI);
#else
	uint32_t W[9];
	for (uint32_t k = 0; k < 9u; k++) W[k] = AW(rec + k);
	uint32_t info = W[8];
#endif
	const uint32_t pos = info & 0xffffu;
	uint32_t mask = (info >> 24) & 0xfu;
	const uint32_t num = 4u - (uint32_t)__builtin_popcount(mask);
	uint32_t chr = 0, last = 0;
	if (num > 1u) {
		/* (range_source.h:341, 417: the total is a TIndexOffU, the dart a 32-bit draw modulo it) */
		bt_row tot = 0;
		for (uint32_t c = 0; c < 4u; c++) if (!((mask >> c) & 1u)) tot += W[4u + c] - W[c];
		uint32_t dart = (uint32_t)(PM_RND(X, d) % tot);
		for (uint32_t c = 0; c < 4u; c++) {
			if ((mask >> c) & 1u) continue;
			const bt_row w = W[4u + c] - W[c];
			chr = c;
			if (c == 3u || dart < w) break;
			dart -= (uint32_t)w;
		}
		mask |= 1u << chr;
		info = (info & ~(0xfu << 24)) | (mask << 24);
	} else {
		last = 1;
		chr = !(mask & 1u) ? 0u : !(mask & 2u) ? 1u : !(mask & 4u) ? 2u : 3u;
		info |= 1u << 28;
	}
	AW(rec + BF_ALI) = info;
	const bt_row top = W[chr], bot = W[4u + chr];
	const uint32_t depth = pos + rdepth;
	const uint32_t d01 = P[BR_D01], d23 = P[BR_D23];
	const uint32_t d0 = d01 & 0xffffu, d1 = d01 >> 16, d2 = d23 & 0xffffu, d3 = d23 >> 16;
	const uint32_t nd0 = depth < d1 ? d1 : d0, nd1 = depth < d2 ? d2 : d1, nd2 = depth < d3 ? d3 : d2;
	const uint32_t hamadd = best & 0x3fffu;
	uint32_t hilo = P[BR_HILO];
	if (depth < depth5) hilo += 1u; else if (depth < seedLen) hilo += 1u << 16;
	const uint32_t bcost = P[BR_COSTHAM] & 0xffffu, bham = P[BR_COSTHAM] >> 16, bned = P[BR_EDIT] >> 16;
	const uint32_t nb = br_new(X, id, nd0 | (nd1 << 16), nd2 | (d3 << 16), depth + 1u, 0, bcost,
	                           (bham + hamadd) & 0xffffu, top, bot, b, depth | (chr << 10) | ((bned + 1u) << 16), hilo);
	uint32_t f = P[BR_FLAGS];
	if (numNotElim == 1u && last) f |= BRF_EXHAUSTED;
	else if (numTied == 1u && last && best != next) {
		f = (f & 0xffffu) | BRF_DELAYED | (((bcost - best + next) & 0xffffu) << 16);
	}
	AW(b + BR_FLAGS) = f;
	return nb;
}

/* PathManager::splitAndPrep (range_source.h:1460-1518); false = the search of this leaf ends now */
BF_FN bool pm_split_and_prep(BfLane& X, uint32_t d, uint32_t seedLen, uint32_t depth5, bool useBtCnt)
{
	if (pm_size(X, d) == 0) return true;
	if (useBtCnt && X.btCnt == 0) return false;
	uint32_t f = pm_front(X, d);
	while (AW(f + BR_FLAGS) & BRF_DELAYED) {
		pm_pop(X, d);
		const uint32_t fl = AW(f + BR_FLAGS);
		br_set_cost(X, f, fl >> 16);
		AW(f + BR_FLAGS) = fl & 0xffffu & ~BRF_DELAYED;
		pm_push(X, d, f);
		f = pm_front(X, d);
		if (X.ovf) return false;
	}
	uint32_t fresh = 0;                                  /* the branch br_split made: br_new stored it prepped */
	if (AW(f + BR_FLAGS) & BRF_CURTAILED) {
		if (useBtCnt) { if (--X.btCnt == 0) return false; }
		const uint32_t nb = br_split(X, d, f, seedLen, depth5);
		if (X.ovf) return false;
		if (AW(f + BR_FLAGS) & BRF_EXHAUSTED) { pm_pop(X, d); pm_free_id(X, d, AW(f + BR_ID)); }
		pm_push(X, d, nb);
		fresh = nb;
	}
	if (pm_size(X, d) && pm_front(X, d) != fresh) br_prep(X, pm_front(X, d));
	return true;
}

/* ---- the leaf: EbwtRangeSourceDriver + EbwtRangeSource ------------------------------------------- */
BF_INL const BfSpec& leaf_spec(BfLane& X, uint32_t d) { return X.P->specs[(AW(d + DR_KIND) >> 16) & 0xffu]; }

BF_INL uint32_t bf_cext(uint32_t cext, uint32_t sRight, uint32_t s, uint32_t len)
{
	return cext == BF_PIN_SEED ? s : cext == BF_PIN_HI_HALF ? sRight : cext == BF_PIN_BEGIN ? 0u : len;
}

BF_FN void leaf_init(BfLane& X, uint32_t d, uint32_t spec)
{
	for (uint32_t k = 0; k < BF_DRW; k++) AW(d + k) = 0;
	AW(d + DR_KIND) = BF_LEAF | ((uint32_t)X.P->specs[spec].fw << 8) | ((uint32_t)X.P->specs[spec].mate << 9) | (spec << 16);
	AW(d + DR_FLAGS) = BF_F_DONE;
}

/* the seed range a partial-alignment extender starts from: the generator's current range, copied
 * (EbwtRangeSource::setQuery keeps a copy, ebwt_search_backtrack.h:1841) */
BF_FN void leaf_take_seed(BfLane& X, uint32_t d, uint32_t src)
{
	const uint32_t n = AW(src + LF_CURCOST) >> 16, sq = AW(src + LF_QLEN) & 0xffffu;
	AW(d + LF_SEED) = (AW(src + LF_CURCOST) & 0xffffu) | (n << 16);
	uint32_t b = AW(src + LF_CURBR);
	/* the edit chain runs from the newest edit to the oldest; the list order is oldest first */
	for (uint32_t k = n; k-- > 0 && b; ) {
		const uint32_t e = AW(b + BR_EDIT);
		if (k < 3u) AW(d + LF_SEEDMM0 + k) = (sq - (e & 0x3ffu) - 1u) | (((e >> 10) & 3u) << 16);
		b = AW(b + BR_PARENT);
	}
	AW(d + LF_RSFLAGS) |= 8u;
}

/* the two lowest quality characters (the lowest twice if it occurs twice) among query offsets qlen-k-1, k in [k0, k1), of
 * the string a leaf reads its penalties from -- a contiguous stretch of the stored row either way round, fetched in
 * 16-byte pieces instead of a character at a time */
BF_FN void bf_qual_low2(const BfRead& R, uint32_t fw, uint32_t ebwtFw, uint32_t qlen, uint32_t k0, uint32_t k1, uint32_t& l1, uint32_t& l2)
{
	l1 = 0xffu; l2 = 0xffu;
	if (k1 <= k0) return;
	const uint32_t lo = (fw == ebwtFw) ? qlen - k1 : R.len - qlen + k0, hi = (fw == ebwtFw) ? qlen - k0 : R.len - qlen + k1;
	for (uint32_t base = lo & ~15u; base < hi; base += 16u) {
		const BtU4 v = bt_ld4((const void*)(R.qual + base));
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		for (uint32_t j = 0; j < 16u; j++) {
			const uint32_t i = base + j;
			if (i < lo || i >= hi) continue;
			const uint32_t c = (w[j >> 2] >> (8u * (j & 3u))) & 0xffu;
			if (c < l1) { l2 = l1; l1 = c; } else if (c < l2) l2 = c;
		}
	}
}

/* SingleRangeSourceDriver::setQueryImpl (range_source.h:1750-1771) with EbwtRangeSource::setQuery
 * (ebwt_search_backtrack.h:1831-1870), EbwtRangeSourceDriver::initRangeSource (:2721-2806) and
 * EbwtRangeSource::initBranch (:1920-2051) */
BF_FN void leaf_set_query(BfLane& X, uint32_t d, uint32_t seedSrc)
{
	BF_PT0(t_setq);
	const BfSpec& sp = leaf_spec(X, d);
	const BtIndexDev& ix = X.ix[sp.mirror];
	const uint32_t maq = X.P->maq;
	AW(d + DR_FLAGS) = 0;
	/* a leaf's query is set once, right after leaf_init zeroed its record (the tree is rebuilt for every read, an extender
	 * is made for every seed hit): its PathManager words are zeros, no need to fetch them */
#if defined(BF_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
	for (uint32_t k = 0; k < 6u; k++) if ((uint32_t)AW(d + LF_HEAP + k) != 0u) { fprintf(stderr, "BF_CHECK: leaf_set_query on a used leaf\n"); abort(); }
#endif
	for (uint32_t k = 0; k < 7u; k++) X.pm[k] = 0;
	pm_reset(X, d);
	const BfRead& R = X.R[sp.mate];
	const uint32_t len = R.len;
	AW(d + LF_RSFLAGS) = 0;
	if (seedSrc) leaf_take_seed(X, d, seedSrc);
	PMW(LF_RND) = R.seed;
	/* initRangeSource */
	const uint32_t s = sp.seedLen > 0 ? (sp.seedLen < len ? sp.seedLen : len) : len;
	uint32_t sRight = s >> 1;
	if ((s & 1u) != 0 && !sp.nudgeLeft) sRight++;
	const uint32_t r0 = bf_cext(sp.rev[0], sRight, s, len), r1 = bf_cext(sp.rev[1], sRight, s, len);
	const uint32_t r2 = bf_cext(sp.rev[2], sRight, s, len), r3 = bf_cext(sp.rev[3], sRight, s, len);
	uint32_t qlen = len;
	if (sp.seed && len > s) qlen = s;
	AW(d + LF_QLEN) = qlen | (len << 16);
	/* the quality string handed to initRangeSource is qual for fw == ebwtFw, qualRev otherwise: the
	 * same string the source reads its penalties from (ebwt_search_backtrack.h:1833-1839) */
	const uint32_t ebwtFw = !sp.mirror;
	uint32_t minCost = 0;
	if (sp.reportExacts) {
	} else if (!sp.halfAndHalf && r0 < s) {
		minCost = 1u << 14;
		uint32_t low = 0xffu;
		{ uint32_t l2; bf_qual_low2(R, sp.fw, ebwtFw, qlen, r0, s, low, l2); }
		minCost += bt_mm_penalty(maq, bf_phred(low));
	} else if (sp.halfAndHalf && sRight > 0 && sRight < (s - 1u)) {
		minCost = (sp.seed ? 3u : 2u) << 14;
		uint32_t low1 = 0xffu;
		uint32_t l21 = 0xffu, l22 = 0xffu;
		{ uint32_t l2; bf_qual_low2(R, sp.fw, ebwtFw, qlen, 0, sRight, low1, l2); }
		bf_qual_low2(R, sp.fw, ebwtFw, qlen, sRight, s, l21, l22);
		minCost += bt_mm_penalty(maq, bf_phred(low1));
		minCost += bt_mm_penalty(maq, bf_phred(l21));
		if (sp.halfAndHalf > 2 && l22 != 0xffu) minCost += bt_mm_penalty(maq, bf_phred(l22));
	}
	minCost &= 0xffffu;
	AW(d + LF_REV01) = r0 | (r1 << 16); AW(d + LF_REV23) = r2 | (r3 << 16); AW(d + LF_D53) = sRight | (s << 16);
	/* initBranch */
	uint32_t rsf = AW(d + LF_RSFLAGS);
	const uint32_t valid = rsf & 8u;
	const uint32_t icost = valid ? (AW(d + LF_SEED) & 0xffffu) : 0u;
	const uint32_t iham = valid ? (icost & 0x3fffu) : 0u;
	bool go = true;
	if (qlen < 4u) {
		uint32_t maxmms = 0;
		if (r0 != r1) maxmms = 1;
		if (r1 != r2) maxmms = 2;
		if (r2 != r3) maxmms = 3;
		if (qlen <= maxmms) { rsf |= 1u | 4u; go = false; }
	}
	/* the leaf's query character at offset i of its (possibly seed-edited) query -- qry_ / qryBuf_
	 * (ebwt_search_backtrack.h:1831-1861) -- with the seed's edits read once instead of at every character */
	uint32_t sqN = 0, sqM[3] = {0, 0, 0};
	if (valid) { sqN = AW(d + LF_SEED) >> 16; for (uint32_t k = 0; k < 3u; k++) sqM[k] = AW(d + LF_SEEDMM0 + k); }
	auto qry = [&](uint32_t i) -> uint32_t {
		uint32_t c = bf_base(R, sp.fw, !sp.mirror, i);
		for (uint32_t k = 0; k < 3u; k++) if (k < sqN && len - (sqM[k] & 0xffffu) - 1u == i) c = sqM[k] >> 16;
		return c;
	};
#define BF_LQ(i) qry(i)
	uint32_t nsInFtab = 0;
	/* a read without an N has none in its seed or its ftab characters (a seed's edits put in reference bases): the two
	 * tallies below are a base fetch per position, each waited for before the next */
	if (go && ((X.hasN >> sp.mate) & 1u)) {
		/* tallyNs (ebwt_search_backtrack.h:2490-2523) */
		uint32_t nsInSeed = 0;
		for (uint32_t i = 0; i < r3 && go; i++) {
			if (BF_LQ(qlen - i - 1u) == 4u) {
				nsInSeed++;
				if (nsInSeed == 1u) { if (i < r0) go = false; }
				else if (nsInSeed == 2u) { if (i < r1) go = false; }
				else if (nsInSeed == 3u) { if (i < r2) go = false; }
				else go = false;
			}
		}
		if (go) for (uint32_t i = 0; i < ix.ftabChars && i < qlen; i++) if (BF_LQ(qlen - i - 1u) == 4u) nsInFtab++;
	}
	if (go) {
		const uint32_t ftabChars = ix.ftabChars;
		const uint32_t m = r0 < qlen ? r0 : qlen;
		const bool skipInvalidExact = !sp.reportExacts && qlen == ftabChars;
		const uint32_t d01 = r0 | (r1 << 16), d23 = r2 | (r3 << 16);
		if (nsInFtab == 0 && m >= ftabChars && !skipInvalidExact) {
			/* calcFtabOff (:2530-2544) on the query's last ftabChars characters taken from two 16-byte pieces of the stored
			 * row (they are next to each other there, whichever way round the leaf reads it) instead of one fetch each */
			uint32_t off;
			{
				const uint32_t a = qlen - ftabChars;                     /* offsets a .. qlen-1 */
				const uint32_t lo = (sp.fw == ebwtFw) ? a : len - qlen, base16 = lo & ~15u;
				const BtU4 v0 = bt_ld4((const void*)(R.seq + base16)), v1 = bt_ld4((const void*)(R.seq + base16 + (lo + ftabChars > base16 + 16u ? 16u : 0u)));
				const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
				auto chr = [&](uint32_t i) -> uint32_t {               /* the query character at i, for i in [a, qlen) */
					const uint32_t sidx = ((sp.fw == ebwtFw) ? i : len - 1u - i) - base16;
					uint32_t c = (w8[sidx >> 2] >> (8u * (sidx & 3u))) & 0xffu;
					if (!sp.fw && c < 4u) c ^= 3u;
					for (uint32_t k = 0; k < 3u; k++) if (k < sqN && len - (sqM[k] & 0xffffu) - 1u == i) c = sqM[k] >> 16;
					return c;
				};
				off = chr(a);
				for (uint32_t i = ftabChars - 1u; i > 0; i--) off = (off << 2) | chr(qlen - i);
			}
			const bt_row top = bt_ftab_hi(ix, off), bot = bt_ftab_lo(ix, off + 1u);
			X.c_ftab++;
			if (qlen == ftabChars && bot > top) {
				AR_SET(d + LF_CURTOP, top); AR_SET(d + LF_CURBOT, bot);
				AW(d + LF_CURCOST) = icost | ((valid ? (AW(d + LF_SEED) >> 16) : 0u) << 16);
				AW(d + LF_CURBR) = 0;
				rsf |= 2u;
			} else if (bot > top) {
				const uint32_t b = br_new(X, pm_alloc_id(X, d), d01, d23, 0, ftabChars, icost, iham, top, bot, 0, 0, 0);
				if (!X.ovf) pm_push(X, d, b);
			}
		} else {
			const uint32_t b = br_new(X, pm_alloc_id(X, d), d01, d23, 0, 0, icost, iham, 0, 0, 0, 0, 0);
			if (!X.ovf) pm_push(X, d, b);
		}
	}
	AW(d + LF_RSFLAGS) = rsf;
	const uint32_t mc = icost > minCost ? icost : minCost;
	AW(d + DR_COST) = mc | (minCost << 16);
	AW(d + DR_FLAGS) = ((rsf & 1u) ? BF_F_DONE : 0u) | ((rsf & 2u) ? BF_F_FOUND : 0u);
	pm_leave(X, d);
	BF_PADD(BP_SETQ, t_setq);
}
#undef BF_LQ

/* EbwtRangeSource::advanceBranch (ebwt_search_backtrack.h:2059-2361), until = ADV_COST_CHANGES, inside
 * SingleRangeSourceDriver::advanceImpl (range_source.h:1777-1838) -- as five pieces over one record of locals, so that the
 * same statements serve the engine run call by call (leaf_advance below: the pieces in their loops) and the wavefront
 * automaton at the end of this file (one piece per round, whichever leaf of whichever read a lane is on).
 *
 * The reference's steps, organised around what a lane waits for.  A branch that is simply extended -- no alternative taken,
 * nothing curtailed -- stays the queue's front with its cost (the queue is not touched, and PathManager::splitAndPrep on
 * such a front only preps it): the next step goes on from the record in registers (la_step), prepped in place, instead of
 * reading the queue, the front's flags and cost and the record again (six dependent waits a base); its base and quality
 * were fetched beside this step's rank loads; and the record's words go back to the arena once, when the streak ends
 * (la_send).  What the arena holds after a streak is what the long way round writes. */
struct BfLeafSt {
	uint32_t d;                   /* the leaf */
	uint32_t spec;                /* its BfSpec */
	uint32_t fl;                  /* its DR_FLAGS when the advance began */
	uint32_t qlen, depth5, depth3;
	uint32_t seedN, seedM[3];     /* the seed's edits the query carries (seedEdits): fixed while the leaf advances */
	uint32_t br, R[BF_BRW];       /* the queue's front and its record */
	uint32_t cost, nedits, rdepth;/* of that branch: fixed while it is extended */
	bt_row   top, bot;
	uint32_t pfC, pfQ;
	bool seedEdits, found, curtail, extended, tbNew, dirty, havePf;
};

/* leaf_advance's prologue; false: the leaf is done, there is nothing to advance (its flags say so now) */
BF_FNI bool la_enter(BfLane& X, BfLeafSt& S, uint32_t d)
{
	const uint32_t fl = AW(d + DR_FLAGS);
	if ((fl & BF_F_DONE) || (AW(d + LF_HEAPSZ) & 0xffffu) == 0) { AW(d + DR_FLAGS) = fl | BF_F_DONE; return false; }
	pm_enter(X, d);
	S.d = d; S.fl = fl; S.spec = (AW(d + DR_KIND) >> 16) & 0xffu;
	S.qlen = AW(d + LF_QLEN) & 0xffffu;
	const uint32_t d53 = AW(d + LF_D53);
	S.depth5 = d53 & 0xffffu; S.depth3 = d53 >> 16;
	S.found = false;
	S.seedEdits = (AW(d + LF_RSFLAGS) & 8u) != 0;
	S.seedN = 0; S.seedM[0] = S.seedM[1] = S.seedM[2] = 0;
	if (S.seedEdits) { S.seedN = AW(d + LF_SEED) >> 16; for (uint32_t k = 0; k < 3u; k++) S.seedM[k] = AW(d + LF_SEEDMM0 + k); }
	return true;
}

/* the queue's front, its record in one go (four independent 16-byte loads, one wait) */
BF_FNI void la_front(BfLane& X, BfLeafSt& S)
{
	BF_PT0(t_front);
	const uint32_t br = pm_front(X, S.d);
	uint32_t* R = S.R;
	{
		const BtU4 r0 = bt_ld4((const void*)(X.A + br)), r1 = bt_ld4((const void*)(X.A + br + 4u));
		const BtU4 r2 = bt_ld4((const void*)(X.A + br + 8u)), r3 = bt_ld4((const void*)(X.A + br + 12u));
		R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y; R[6] = r1.z; R[7] = r1.w;
		R[8] = r2.x; R[9] = r2.y; R[10] = r2.z; R[11] = r2.w; R[12] = r3.x; R[13] = r3.y; R[14] = r3.z; R[15] = r3.w;
#if BT_WIDE
		const BtU4 r4 = bt_ld4((const void*)(X.A + br + 16u));
		R[16] = r4.x; R[17] = r4.y; R[18] = r4.z; R[19] = r4.w;
#endif
	}
	S.br = br;
	S.cost = R[BR_COSTHAM] & 0xffffu;
	S.nedits = R[BR_EDIT] >> 16;
	S.rdepth = R[BR_RDLEN] & 0xffffu;
	S.top = S.bot = 0;
	S.curtail = S.extended = S.tbNew = S.dirty = S.havePf = false;
	S.pfC = S.pfQ = 0;
	BF_PADD(BP_FRONT, t_front);
}

/* one step of the streak; true: the branch was simply extended and the next step goes on from the registers */
BF_FNI bool la_step(BfLane& X, BfLeafSt& S)
{
	const BfSpec& sp = X.P->specs[S.spec];
	const BtIndexDev& ix = X.ix[sp.mirror];
	const uint32_t d = S.d, br = S.br, qlen = S.qlen, depth5 = S.depth5, depth3 = S.depth3;
	const uint32_t maq = X.P->maq;
	const BfRead& RD = X.R[sp.mate];
	const uint32_t ebwtFw = !sp.mirror;
	uint32_t* R = S.R;
	const uint32_t cost = S.cost, nedits = S.nedits, rdepth = S.rdepth;
	const uint32_t blen = R[BR_RDLEN] >> 16;
	const uint32_t depth = rdepth + blen;
	uint32_t cur = 0;
	bt_row top = RR(R, BR_TOP), bot = RR(R, BR_BOT);
	bool curtail = false, extended = false, tbNew = false;
	bool hit = false;
	/* hhCheckTop (:2444-2475) */
	if (sp.halfAndHalf && ((depth == depth5 && nedits == 0) || (depth == depth3 && nedits < sp.halfAndHalf))) {
		curtail = true;
	} else {
		cur = qlen - depth - 1u;
		if (depth < qlen) {
			uint32_t c = S.havePf ? S.pfC : bf_base(RD, sp.fw, ebwtFw, cur);
			if (S.seedEdits) {
				const uint32_t full = RD.len;                          /* the seed's edits override the read's characters (qryBuf_), from registers */
				for (uint32_t k = 0; k < 3u; k++) if (k < S.seedN && full - (S.seedM[k] & 0xffffu) - 1u == cur) c = S.seedM[k] >> 16;
			}
			const uint32_t q = bt_mm_penalty(maq, bf_phred(S.havePf ? S.pfQ : bf_qualc(RD, sp.fw, ebwtFw, cur)));
			/* the next position's base and quality, in flight beside this step's rank loads */
			S.havePf = cur > 0;
			if (S.havePf) { S.pfC = bf_base(RD, sp.fw, ebwtFw, cur - 1u); S.pfQ = bf_qualc(RD, sp.fw, ebwtFw, cur - 1u); }
			const uint32_t ham = R[BR_COSTHAM] >> 16;
			const uint32_t d0 = R[BR_D01] & 0xffffu;
			const bool alt = depth >= d0 && ham + q <= sp.qualLim;
			bt_row otop = top;
			if (c == 4u && depth > 0) top = bot = 1;
			const uint32_t fl = R[BR_FLAGS];
			bt_row tops[4] = {0, 0, 0, 0}, bots[4] = {0, 0, 0, 0};
			bool ranges = false;
			/* which LF-mapping the step needs is decided first, the rows' rank blocks are fetched at ONE place in the code --
			 * the lanes of a wavefront that are stepping, whatever their cases, wait for them once -- and the case is
			 * finished afterwards: 1 = the four ranges of both rows (countFwSideEx x 2), 2 = mapLF1(otop, ltop_)
			 * (ebwt.h:2530-2560), 3 = mapLF1(top_, ltop_, c) (ebwt.h:2494-2524), 4 = mapLF of both rows for c */
			uint32_t lfCase = 0;
			bool altCase = false;
			if (top == 0 && bot == 0) {
				tops[0] = ix.fchr[0]; bots[0] = tops[1] = ix.fchr[1]; bots[1] = tops[2] = ix.fchr[2];
				bots[2] = tops[3] = ix.fchr[3]; bots[3] = ix.fchr[4];
				ranges = true;
				if (c < 4u) { top = tops[c]; bot = bots[c]; }
			} else if (alt && (bot > top || c == 4u)) {
				if (fl & BRF_LBOT) lfCase = 1u;
				else { X.c_lf1++; if (otop != ix.zOff) lfCase = 2u; }
				ranges = true; altCase = true;
			} else if (bot > top) {
				if (c < 4u) lfCase = (top + 1u == bot) ? 3u : 4u;
			}
			if (lfCase) {
				bt_row la[4], lb[4] = {0, 0, 0, 0}; uint32_t LA, LB;
				const bt_row ra = RR(R, BR_LTOP), rb = RR(R, BR_LBOT);
				const bool two = lfCase == 1u || lfCase == 4u;
				bt_rank4(ix, ra, la, &LA);
				if (two) bt_rank4(ix, rb, lb, &LB);
				if (lfCase == 1u) {
					for (uint32_t k = 0; k < 4u; k++) { tops[k] = la[k]; bots[k] = lb[k]; }
					X.c_lfex++; if (ra / 448u == rb / 448u) X.c_same++;
				} else if (lfCase == 2u) {
					otop = la[LA];
					tops[LA] = otop; bots[LA] = otop + 1u;
				} else if (lfCase == 3u) {
					X.c_lf1++;
					if (LA != c || top == ix.zOff) top = bot = BT_OFF_MASK;
					else { top = la[c]; bot = top + 1u; }
				} else {
					X.c_lf2++; if (ra / 448u == rb / 448u) X.c_same++;
					top = la[c]; bot = lb[c];
				}
			}
			if (altCase) { if (c < 4u) { top = tops[c]; bot = bots[c]; } else top = bot = 1; }
			if (ranges) {
				/* Branch::installRanges (range_source.h:970-1023): a record only for a position that
				 * is a legitimate place to branch from and still has an untried substitution */
				uint32_t mask = 0xfu;
				if (q <= sp.qualLim - ham) {
					for (uint32_t k = 0; k < 4u; k++) if (c != k && bots[k] > tops[k]) mask &= ~(1u << k);
				}
				if (mask != 0xfu && depth >= d0) {
					const uint32_t nalt = R[BR_NALT];
					const uint32_t rec = bf_alloc(X, BF_ALW);
					if (nalt == 0) { AW(br + BR_ALT) = rec; X.growing = br; }
					else if (X.growing != br || rec != R[BR_ALT] + nalt * BF_ALW) X.ovf = 2;   /* contiguity broken: a bug */
					if (!X.ovf) {
						for (uint32_t k = 0; k < 4u; k++) { AR_SET(rec + k * BF_RW, tops[k]); AR_SET(rec + (4u + k) * BF_RW, bots[k]); }
						AW(rec + BF_ALI) = blen | (q << 16) | (mask << 24);
						AW(br + BR_NALT) = nalt + 1u;
						if (nalt == 0) R[BR_ALT] = rec;
						R[BR_NALT] = nalt + 1u;
					}
				}
			}
		} else {
			cur = 0;
		}
		tbNew = true;                                  /* top and bot are written when the streak ends (la_send) */
		const bool empty = top == bot;
		hit = cur == 0 && !empty;
		const bool invalidExact = hit && nedits == 0 && !sp.reportExacts;
		/* hhCheck (:2397-2436) */
		bool hhOk = true;
		if (sp.halfAndHalf) {
			if (depth == depth5 - 1u && !empty) hhOk = nedits > 0;
			else if (depth == depth3 - 1u && !empty) {
				const uint32_t hilo = R[BR_HILO];
				hhOk = nedits >= sp.halfAndHalf && (hilo & 0xffffu) != 0 && (hilo >> 16) != 0;
			}
		}
		if (!hhOk) { curtail = true; hit = false; }
		else if (hit && !invalidExact) {
			AR_SET(d + LF_CURTOP, top); AR_SET(d + LF_CURBOT, bot);
			AW(d + LF_CURCOST) = cost | ((nedits + (S.seedEdits ? S.seedN : 0u)) << 16);
			AW(d + LF_CURBR) = br;
			S.found = true;
			curtail = true;
		} else if (empty || cur == 0) curtail = true;
		else { R[BR_RDLEN] = rdepth | ((blen + 1u) << 16); extended = true; }   /* Branch::extend */
	}
	S.top = top; S.bot = bot; S.curtail = curtail; S.extended = extended; S.tbNew = tbNew;
	if (!(extended && !X.ovf && (R[BR_FLAGS] & (BRF_DELAYED | BRF_CURTAILED)) == 0 && !(sp.useBtCnt != 0 && X.btCnt == 0))) return false;
	/* splitAndPrep on an untouched queue whose front is this branch does one thing, prep: done on the registers */
	uint32_t f = R[BR_FLAGS];
	if (bot > top + 1u) { f |= BRF_LTOP | BRF_LBOT; RR_SET(R, BR_LTOP, top); RR_SET(R, BR_LBOT, bot); }
	else if (bot > top) { f = (f | BRF_LTOP) & ~BRF_LBOT; RR_SET(R, BR_LTOP, top); }
	R[BR_FLAGS] = f; RR_SET(R, BR_TOP, top); RR_SET(R, BR_BOT, bot);
	S.dirty = true;
#if defined(BF_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
	if ((uint32_t)AW(PMW(LF_HEAP)) != br || X.pm[6] != br || PMW(LF_PMCOST) != cost) { fprintf(stderr, "BF_CHECK: the extended branch is not the queue's front\n"); abort(); }
#endif
	return true;
}

/* the streak has ended: the arena gets what its steps would have written one by one, before anything reads it; curtail,
 * splitAndPrep.  true: the leaf's advance goes on with the queue's new front (la_front), false: it is over (la_exit) */
BF_FNI bool la_send(BfLane& X, BfLeafSt& S)
{
	const BfSpec& sp = X.P->specs[S.spec];
	const uint32_t d = S.d, br = S.br;
	const uint32_t* R = S.R;
	if (S.tbNew) { AR_SET(br + BR_TOP, S.top); AR_SET(br + BR_BOT, S.bot); }
	else if (S.dirty) { AR_SET(br + BR_TOP, RR(R, BR_TOP)); AR_SET(br + BR_BOT, RR(R, BR_BOT)); }
	if (S.dirty) { AW(br + BR_FLAGS) = R[BR_FLAGS]; AR_SET(br + BR_LTOP, RR(R, BR_LTOP)); AR_SET(br + BR_LBOT, RR(R, BR_LBOT)); }
	if (S.dirty || S.extended) AW(br + BR_RDLEN) = R[BR_RDLEN];
	{ BF_PT0(t_curtail); if (S.curtail) pm_curtail_regs(X, d, br, S.depth3, R); BF_PADD(BP_CURTAIL, t_curtail); }
	if (X.ovf) return false;
	{ BF_PT0(t_split); const bool sp_ok = pm_split_and_prep(X, d, S.depth3, S.depth5, sp.useBtCnt != 0); BF_PADD(BP_SPLIT, t_split); if (!sp_ok) pm_reset(X, d); }
	if (X.ovf) return false;
	if (pm_size(X, d) == 0) return false;
	/* the queue's cost word is its front's cost: every change of a queued branch's cost (curtail, a delayed increase) is
	 * followed by a pop and a push, which set it */
#if defined(BF_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
	if (PMW(LF_PMCOST) != br_cost(X, AW(PMW(LF_HEAP))) || X.pm[6] != (uint32_t)AW(PMW(LF_HEAP))) { fprintf(stderr, "BF_CHECK: queue cost / front cache out of step\n"); abort(); }
#endif
	if (PMW(LF_PMCOST) != S.cost) return false;
	return !S.found;
}

/* leaf_advance's epilogue: the range source's and the driver's flags and cost */
BF_FNI void la_exit(BfLane& X, BfLeafSt& S)
{
	const uint32_t d = S.d;
	const uint32_t rsf = (AW(d + LF_RSFLAGS) & ~2u) | (S.found ? 2u : 0u);
	AW(d + LF_RSFLAGS) = rsf;
	uint32_t fl = S.fl & ~(BF_F_DONE | BF_F_FOUND);
	if (pm_size(X, d) == 0) fl |= BF_F_DONE;
	const uint32_t pmc = PMW(LF_PMCOST), adj = AW(d + DR_COST) >> 16;
	if (pmc != 0) AW(d + DR_COST) = (pmc > adj ? pmc : adj) | (adj << 16);
	if (rsf & 2u) fl |= BF_F_FOUND;
	AW(d + DR_FLAGS) = fl;
	pm_leave(X, d);
}

/* SingleRangeSourceDriver::advanceImpl (range_source.h:1777-1838): the pieces in their loops */
BF_FN void leaf_advance(BfLane& X, uint32_t d)
{
	BfLeafSt S;
	if (!la_enter(X, S, d)) return;
	BF_PT0(t_leaf);
	do {
		la_front(X, S);
		BF_PT0(t_streak);
		while (la_step(X, S)) { }
		BF_PADD(BP_STREAK, t_streak);
	} while (la_send(X, S));
	BF_PADD(BP_LEAF, t_leaf);
	la_exit(X, S);
}

/* ---- the inner drivers ------------------------------------------------------------------------ */
BF_INL uint32_t dr_kind(BfLane& X, uint32_t d) { return AW(d + DR_KIND) & 0xffu; }
BF_INL uint32_t dr_fw(BfLane& X, uint32_t d) { return (AW(d + DR_KIND) >> 8) & 1u; }
BF_INL uint32_t dr_mate(BfLane& X, uint32_t d) { return (AW(d + DR_KIND) >> 9) & 1u; }
BF_INL uint32_t dr_mincost(BfLane& X, uint32_t d) { return AW(d + DR_COST) & 0xffffu; }
BF_INL void dr_set_mincost(BfLane& X, uint32_t d, uint32_t c) { AW(d + DR_COST) = (AW(d + DR_COST) & 0xffff0000u) | (c & 0xffffu); }
BF_INL bool dr_done(BfLane& X, uint32_t d) { return (AW(d + DR_FLAGS) & BF_F_DONE) != 0; }
BF_INL bool dr_found(BfLane& X, uint32_t d) { return (AW(d + DR_FLAGS) & BF_F_FOUND) != 0; }
BF_INL void dr_set(BfLane& X, uint32_t d, uint32_t bit, bool v) { uint32_t f = AW(d + DR_FLAGS); AW(d + DR_FLAGS) = v ? (f | bit) : (f & ~bit); }

BF_FN void cost_init(BfLane& X, uint32_t d, uint32_t strandFix, uint32_t cap)
{
	for (uint32_t k = 0; k < BF_DRW; k++) AW(d + k) = 0;
	AW(d + DR_KIND) = BF_COST | (1u << 8);
	AW(d + CA_OPTS) = strandFix ? 1u : 0u;
	if (cap) { AW(d + CA_RSS) = bf_alloc(X, cap); AW(d + CA_ACT) = bf_alloc(X, cap); AW(d + CA_NRSS) = cap << 16; }
}
BF_FN void cost_add_rss(BfLane& X, uint32_t d, uint32_t p)
{
	uint32_t n = AW(d + CA_NRSS) & 0xffffu, cap = AW(d + CA_NRSS) >> 16;
	if (n == cap) {
		const uint32_t ncap = cap ? cap * 2u : 4u;
		const uint32_t nr = bf_alloc(X, ncap), na = bf_alloc(X, ncap);
		if (X.ovf) return;
		const uint32_t nact = AW(d + CA_NACT);
		for (uint32_t k = 0; k < n; k++) AW(nr + k) = AW(AW(d + CA_RSS) + k);
		for (uint32_t k = 0; k < nact; k++) AW(na + k) = AW(AW(d + CA_ACT) + k);
		AW(d + CA_RSS) = nr; AW(d + CA_ACT) = na; cap = ncap;
	}
	AW(AW(d + CA_RSS) + n) = p;
	AW(d + CA_NRSS) = (n + 1u) | (cap << 16);
}

/* CostAwareRangeSourceDriver::sortActives (range_source.h:2370-2415) */
BF_FN void cost_sort_actives_body(BfLane& X, uint32_t d);
BF_FN void cost_sort_actives(BfLane& X, uint32_t d) { BF_PT0(t_sort); cost_sort_actives_body(X, d); BF_PADD(BP_SORT, t_sort); }
BF_FN void cost_sort_actives_body(BfLane& X, uint32_t d)
{
	const uint32_t vec = AW(d + CA_ACT);
	uint32_t n = AW(d + CA_NACT), sz = n;
	if (n <= 16u) {
		/* the same selection sort (and the same draws) on a copy of what it looks at: every child's flags and cost are
		 * fetched once, side by side, instead of inside the two loops, each fetch waited for */
		uint32_t v[16], fl[16], co[16];
		BF_UNROLL16 for (uint32_t i = 0; i < 16u; i++) v[i] = i < n ? (uint32_t)AW(vec + i) : 0u;
		BF_UNROLL16 for (uint32_t i = 0; i < 16u; i++) { fl[i] = i < n ? (uint32_t)AW(v[i] + DR_FLAGS) : 0u; co[i] = i < n ? ((uint32_t)AW(v[i] + DR_COST) & 0xffffu) : 0u; }
		const uint32_t n0 = n;
		uint32_t rs = AW(d + CA_RND);
		bool drew = false;
		for (uint32_t i = 0; i < sz;) {
			if ((fl[i] & BF_F_DONE) && !(fl[i] & BF_F_FOUND)) {
				for (uint32_t k = i; k + 1u < n; k++) { v[k] = v[k + 1u]; fl[k] = fl[k + 1u]; co[k] = co[k + 1u]; }
				n--; sz--;
				continue;
			}
			uint32_t minCost = co[i], minOff = i;
			for (uint32_t j = i + 1u; j < sz; j++) {
				if ((fl[j] & BF_F_DONE) && !(fl[j] & BF_F_FOUND)) continue;
				const uint32_t cj = co[j];
				if (cj < minCost) { minCost = cj; minOff = j; }
				else if (cj == minCost) { drew = true; if (bf_rnd(rs) & 0x1000u) minOff = j; }
			}
			if (i != minOff) {
				uint32_t t = v[i]; v[i] = v[minOff]; v[minOff] = t;
				t = fl[i]; fl[i] = fl[minOff]; fl[minOff] = t;
				t = co[i]; co[i] = co[minOff]; co[minOff] = t;
			}
			i++;
		}
		for (uint32_t k = 0; k < n0; k++) AW(vec + k) = v[k];          /* the slots past n included: what the in-place shifts leave there */
		if (drew) AW(d + CA_RND) = rs;
		AW(d + CA_NACT) = n;
		if (AW(d + CA_DELAYED) == 0 && sz > 0) dr_set_mincost(X, d, co[0]);
		return;
	}
	{
		/* more children than that (a seeded driver with hundreds of extenders: the reads that take longest): the same sort
		 * on a key array in the arena beside the vector -- cost | dead << 16 per child, gathered once, eight at a time --
		 * so that the inner loop reads one contiguous word per child instead of the child's offset, then its flags, then
		 * its cost */
		uint32_t key = AW(d + CA_KEY), kcap = AW(d + CA_KEYCAP);
		if (kcap < n) {
			const uint32_t ncap = (n + n / 2u + 15u) & ~7u;
			key = bf_alloc(X, ncap);
			if (X.ovf) return;                                 /* the read is on its way to a larger arena */
			AW(d + CA_KEY) = key; AW(d + CA_KEYCAP) = ncap;
		}
		for (uint32_t i0 = 0; i0 < n; i0 += 8u) {
			uint32_t v8[8], f8[8], c8[8];
			for (uint32_t j = 0; j < 8u; j++) v8[j] = i0 + j < n ? (uint32_t)AW(vec + i0 + j) : 0u;
			for (uint32_t j = 0; j < 8u; j++) { f8[j] = i0 + j < n ? (uint32_t)AW(v8[j] + DR_FLAGS) : 0u; c8[j] = i0 + j < n ? (uint32_t)AW(v8[j] + DR_COST) : 0u; }
			for (uint32_t j = 0; j < 8u; j++) if (i0 + j < n) AW(key + i0 + j) = (c8[j] & 0xffffu) | (((f8[j] & BF_F_DONE) && !(f8[j] & BF_F_FOUND)) ? 0x10000u : 0u);
		}
		uint32_t rs = AW(d + CA_RND);
		bool drew = false;
		for (uint32_t i = 0; i < sz;) {
			const uint32_t ki = AW(key + i);
			if (ki >> 16) {
				for (uint32_t k = i; k + 1u < n; k++) { AW(vec + k) = AW(vec + k + 1u); AW(key + k) = AW(key + k + 1u); }
				n--; sz--;
				continue;
			}
			uint32_t minCost = ki & 0xffffu, minOff = i;
			for (uint32_t j0 = i + 1u; j0 < sz; j0 += 8u) {
				uint32_t k8[8];
				for (uint32_t j = 0; j < 8u; j++) k8[j] = j0 + j < sz ? (uint32_t)AW(key + j0 + j) : 0x10000u;
				for (uint32_t j = 0; j < 8u; j++) {
					if (k8[j] >> 16) continue;
					const uint32_t cj = k8[j] & 0xffffu;
					if (cj < minCost) { minCost = cj; minOff = j0 + j; }
					else if (cj == minCost) { drew = true; if (bf_rnd(rs) & 0x1000u) minOff = j0 + j; }
				}
			}
			if (i != minOff) {
				uint32_t t = AW(vec + i); AW(vec + i) = AW(vec + minOff); AW(vec + minOff) = t;
				t = AW(key + i); AW(key + i) = AW(key + minOff); AW(key + minOff) = t;
			}
			i++;
		}
		if (drew) AW(d + CA_RND) = rs;
		AW(d + CA_NACT) = n;
		if (AW(d + CA_DELAYED) == 0 && sz > 0) dr_set_mincost(X, d, (uint32_t)AW(key) & 0xffffu);
		return;
	}
	for (uint32_t i = 0; i < sz;) {
		const uint32_t vi = AW(vec + i);
		if (dr_done(X, vi) && !dr_found(X, vi)) {
			for (uint32_t k = i; k + 1u < n; k++) AW(vec + k) = AW(vec + k + 1u);
			n--; sz--;
			continue;
		}
		uint32_t minCost = dr_mincost(X, vi), minOff = i;
		for (uint32_t j = i + 1u; j < sz; j++) {
			const uint32_t vj = AW(vec + j);
			if (dr_done(X, vj) && !dr_found(X, vj)) continue;
			const uint32_t cj = dr_mincost(X, vj);
			if (cj < minCost) { minCost = cj; minOff = j; }
			else if (cj == minCost) { if (bf_rnd_at(X, d + CA_RND) & 0x1000u) minOff = j; }
		}
		if (i != minOff) { const uint32_t t = AW(vec + i); AW(vec + i) = AW(vec + minOff); AW(vec + minOff) = t; }
		i++;
	}
	AW(d + CA_NACT) = n;
	if (AW(d + CA_DELAYED) == 0 && sz > 0) dr_set_mincost(X, d, dr_mincost(X, AW(vec)));
}

/* LEVEL 0 = the aligner's driver (children: leaves and seeded pairs), LEVEL 1 = a seeded pair's
 * rsFull_ (children: leaves only).  Two instantiations instead of the reference's virtual recursion. */
template <int LEVEL> BF_FN void child_set_query(BfLane& X, uint32_t d, uint32_t seedSrc);
template <int LEVEL> BF_FN void child_advance(BfLane& X, uint32_t d);
BF_FN uint32_t child_range(BfLane& X, uint32_t d);

/* setQueryImpl (range_source.h:2076-2093) */
template <int LEVEL> BF_FN void cost_set_query(BfLane& X, uint32_t d)
{
	AW(d + DR_FLAGS) = 0; AW(d + CA_LAST) = 0; AW(d + CA_DELAYED) = 0;
	AW(d + CA_OPTS) |= 2u;
	AW(d + CA_RND) = X.R[0].seed;                             /* patsrc->bufa().seed */
	const uint32_t n = AW(d + CA_NRSS) & 0xffffu;
	if (n == 0) return;
	{
		/* the two vectors' offsets once, each child's offset once (setting a child's query does not touch the vectors) */
		const uint32_t rss = AW(d + CA_RSS), act = AW(d + CA_ACT);
		for (uint32_t i = 0; i < n; i++) { const uint32_t c = AW(rss + i); child_set_query<LEVEL>(X, c, 0); AW(act + i) = c; }
	}
	AW(d + CA_NACT) = n;
	dr_set_mincost(X, d, 0);
	cost_sort_actives(X, d);
}

/* foundFirstRange (range_source.h:2311-2362); rss_[i] -- not active_[i] -- supplies mate1()/fw() */
template <int LEVEL> BF_FN bool cost_found_first_range(BfLane& X, uint32_t d, uint32_t r, uint32_t rfw, uint32_t rmate)
{
	dr_set(X, d, BF_F_FOUND, true);
	AW(d + CA_LAST) = r;
	if (AW(d + CA_OPTS) & 1u) {
		const uint32_t sz = AW(d + CA_NACT);
		const uint32_t rcost = AW(r + LF_CURCOST) & 0xffffu;
		for (uint32_t i = 1; i < sz; i++) {
			if (dr_mate(X, AW(AW(d + CA_RSS) + i)) == rmate && dr_fw(X, AW(AW(d + CA_RSS) + i)) != rfw) {
				const uint32_t p = AW(AW(d + CA_ACT) + i);
				const uint32_t mine = dr_mincost(X, d), theirs = dr_mincost(X, p);
				const uint32_t minCost = mine > theirs ? mine : theirs;
				if (minCost > rcost) break;
				while (!dr_done(X, p) && !dr_found(X, p) && !X.ovf) {
					child_advance<LEVEL>(X, p);
					if (dr_mincost(X, p) > minCost) break;
				}
				if (dr_found(X, p)) {
					uint32_t del = child_range(X, p), lastR = r;
					const bt_row wd = AR(del + LF_CURBOT) - AR(del + LF_CURTOP), wl = AR(lastR + LF_CURBOT) - AR(lastR + LF_CURTOP);
					const uint64_t tot = (uint64_t)wd + wl;
					const uint32_t rq = (uint32_t)((uint64_t)bf_rnd_at(X, d + CA_RND) % tot);
					if (rq < wd) { const uint32_t t = lastR; lastR = del; del = t; }
					AW(d + CA_LAST) = lastR; AW(d + CA_DELAYED) = del;
					dr_set(X, p, BF_F_FOUND, false);
				}
				return true;
			}
		}
	}
	return false;
}

/* mateEliminated (range_source.h:2266-2280): only the aligner's own driver mixes mates */
template <int LEVEL> BF_FN bool cost_mate_eliminated(BfLane& X, uint32_t d)
{
	if (LEVEL != 0 || !X.P->paired || BF_IS_V1(*X.P)) return false;      /* V1's drivers hold one mate each */
	const uint32_t n = AW(d + CA_NACT);
	bool m1 = false, m2 = false;
	if (n <= 16u) {
		const uint32_t vec = AW(d + CA_ACT);
		uint32_t a[16], fl[16], kd[16];
		BF_UNROLL16 for (uint32_t i = 0; i < 16u; i++) a[i] = i < n ? (uint32_t)AW(vec + i) : 0u;
		BF_UNROLL16 for (uint32_t i = 0; i < 16u; i++) { fl[i] = i < n ? (uint32_t)AW(a[i] + DR_FLAGS) : BF_F_DONE; kd[i] = i < n ? (uint32_t)AW(a[i] + DR_KIND) : 0u; }
		BF_UNROLL16 for (uint32_t i = 0; i < 16u; i++) if (i < n && !(fl[i] & BF_F_DONE)) { if ((kd[i] >> 9) & 1u) m2 = true; else m1 = true; }
		return !m1 || !m2;
	}
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t a = AW(AW(d + CA_ACT) + i);
		if (!dr_done(X, a)) { if (dr_mate(X, a)) m2 = true; else m1 = true; }
	}
	return !m1 || !m2;
}

/* advanceImpl (range_source.h:2157-2210) */
template <int LEVEL> BF_FN void cost_advance(BfLane& X, uint32_t d)
{
	AW(d + CA_LAST) = 0;
	const uint32_t actSz = AW(d + CA_NACT);
	if (AW(d + CA_DELAYED)) {
		AW(d + CA_LAST) = AW(d + CA_DELAYED); AW(d + CA_DELAYED) = 0;
		dr_set(X, d, BF_F_FOUND, true);
		if (actSz > 0) { const uint32_t a0 = dr_mincost(X, AW(AW(d + CA_ACT))); if (a0 > dr_mincost(X, d)) dr_set_mincost(X, d, a0); }
		else dr_set(X, d, BF_F_DONE, true);
		return;
	}
	if (cost_mate_eliminated<LEVEL>(X, d) || actSz == 0) { AW(d + CA_NACT) = 0; dr_set(X, d, BF_F_DONE, true); return; }
	const uint32_t p = AW(AW(d + CA_ACT));
	const uint32_t precost = dr_mincost(X, p);
	if (!dr_found(X, p)) child_advance<LEVEL>(X, p);
	bool needsSort = false;
	if (dr_found(X, p)) {
		const uint32_t r = child_range(X, p);
		needsSort = cost_found_first_range<LEVEL>(X, d, r, dr_fw(X, p), dr_mate(X, p));
		dr_set(X, p, BF_F_FOUND, false);
	}
	if (dr_done(X, p) || precost != dr_mincost(X, p) || needsSort) {
		cost_sort_actives(X, d);
		if (cost_mate_eliminated<LEVEL>(X, d) || AW(d + CA_NACT) == 0) { AW(d + CA_NACT) = 0; dr_set(X, d, BF_F_DONE, AW(d + CA_DELAYED) == 0); }
	}
}

/* EbwtSeededRangeSourceDriver::setQueryImpl (ebwt_search_backtrack.h:2963-2978) */
BF_FN void seeded_set_query(BfLane& X, uint32_t d)
{
	const uint32_t seed = AW(d + SD_SEED), full = AW(d + SD_FULL);
	AW(d + DR_FLAGS) = 0;
	leaf_set_query(X, seed, 0);
	const uint32_t sadj = AW(seed + DR_COST) >> 16, smin = dr_mincost(X, seed);
	const uint32_t adj = sadj > smin ? sadj : smin;
	AW(d + DR_COST) = adj | (adj << 16);
	/* rsFull_.clearSources(); rsFull_.setQuery() */
	AW(full + CA_NRSS) &= 0xffff0000u; AW(full + CA_NACT) = 0;
	cost_set_query<1>(X, full);
	dr_set_mincost(X, full, adj);
}

/* EbwtSeededRangeSourceDriver::advanceImpl (ebwt_search_backtrack.h:3011-3103) */
BF_FN void seeded_advance(BfLane& X, uint32_t d)
{
	const uint32_t seed = AW(d + SD_SEED), full = AW(d + SD_FULL);
	if (dr_done(X, seed) && dr_done(X, full) && !dr_found(X, seed) && !dr_found(X, full)) { dr_set(X, d, BF_F_DONE, true); return; }
	if (dr_done(X, seed) && !dr_found(X, seed)) {
		dr_set_mincost(X, seed, 0xffffu);
		if (dr_mincost(X, full) > dr_mincost(X, d)) { dr_set_mincost(X, d, dr_mincost(X, full)); return; }
	}
	if (dr_done(X, full) && !dr_found(X, full)) {
		dr_set_mincost(X, full, 0xffffu);
		if (dr_mincost(X, seed) > dr_mincost(X, d)) { dr_set_mincost(X, d, dr_mincost(X, seed)); return; }
	}
	if (dr_mincost(X, full) > dr_mincost(X, seed)) {
		if (!dr_found(X, seed)) leaf_advance(X, seed);
		if (dr_found(X, seed)) {
			dr_set(X, seed, BF_F_FOUND, false);
			const uint32_t scost = AW(seed + LF_CURCOST) & 0xffffu;
			AW(d + DR_COST) = dr_mincost(X, d) | (scost << 16);
			/* rsFact_->create(); rsFull_.addSource(partial, seedRange_) (range_source.h:2098-2111) */
			const uint32_t part = bf_alloc(X, BF_DRW);
			if (X.ovf) return;
			leaf_init(X, part, AW(d + SD_FACT));
			AW(full + CA_LAST) = 0; AW(full + CA_DELAYED) = 0; dr_set(X, full, BF_F_DONE, false);
			leaf_set_query(X, part, seed);
			cost_add_rss(X, full, part);
			if (X.ovf) return;
			const uint32_t na = AW(full + CA_NACT);
			AW(AW(full + CA_ACT) + na) = part; AW(full + CA_NACT) = na + 1u;
			dr_set_mincost(X, full, 0);
			cost_sort_actives(X, full);
			if (dr_found(X, full)) { dr_set(X, d, BF_F_FOUND, true); dr_set(X, full, BF_F_FOUND, false); }
		}
		if (dr_mincost(X, seed) > dr_mincost(X, d)) {
			uint32_t mc = dr_mincost(X, seed);
			if (!dr_done(X, full) && dr_mincost(X, full) < mc) mc = dr_mincost(X, full);
			dr_set_mincost(X, d, mc);
		}
	} else {
		const uint32_t old = dr_mincost(X, full);
		if (!dr_found(X, full)) cost_advance<1>(X, full);
		if (dr_found(X, full)) { dr_set(X, d, BF_F_FOUND, true); dr_set(X, full, BF_F_FOUND, false); }
		if (dr_mincost(X, full) > old) {
			const uint32_t a = dr_mincost(X, full), b = dr_mincost(X, seed);
			dr_set_mincost(X, d, a < b ? a : b);
		}
	}
}

template <int LEVEL> BF_FN void child_set_query(BfLane& X, uint32_t d, uint32_t seedSrc)
{
	if (LEVEL == 0 && dr_kind(X, d) == BF_SEEDED) seeded_set_query(X, d);
	else leaf_set_query(X, d, seedSrc);
}
template <int LEVEL> BF_FN void child_advance(BfLane& X, uint32_t d)
{
	if (LEVEL == 0 && dr_kind(X, d) == BF_SEEDED) seeded_advance(X, d);
	else leaf_advance(X, d);
}
/* RangeSourceDriver::range(): the leaf whose current range it is */
BF_FN uint32_t child_range(BfLane& X, uint32_t d)
{
	if (dr_kind(X, d) == BF_SEEDED) return AW(AW(d + SD_FULL) + CA_LAST);
	return d;
}

/* cost_advance from the return of child_advance on: foundFirstRange, sortActives (range_source.h:2186-2210) */
template <int LEVEL> BF_FN void cost_advance_post(BfLane& X, uint32_t d, uint32_t p, uint32_t precost)
{
	bool needsSort = false;
	if (dr_found(X, p)) {
		const uint32_t r = child_range(X, p);
		needsSort = cost_found_first_range<LEVEL>(X, d, r, dr_fw(X, p), dr_mate(X, p));
		dr_set(X, p, BF_F_FOUND, false);
	}
	if (dr_done(X, p) || precost != dr_mincost(X, p) || needsSort) {
		cost_sort_actives(X, d);
		if (cost_mate_eliminated<LEVEL>(X, d) || AW(d + CA_NACT) == 0) { AW(d + CA_NACT) = 0; dr_set(X, d, BF_F_DONE, AW(d + CA_DELAYED) == 0); }
	}
}
/* seeded_advance from the return of leaf_advance(seed) on (ebwt_search_backtrack.h:3049-3083) */
BF_FN void seeded_post_seed(BfLane& X, uint32_t d, uint32_t seed, uint32_t full)
{
	if (dr_found(X, seed)) {
		dr_set(X, seed, BF_F_FOUND, false);
		const uint32_t scost = AW(seed + LF_CURCOST) & 0xffffu;
		AW(d + DR_COST) = dr_mincost(X, d) | (scost << 16);
		const uint32_t part = bf_alloc(X, BF_DRW);
		if (X.ovf) return;
		leaf_init(X, part, AW(d + SD_FACT));
		AW(full + CA_LAST) = 0; AW(full + CA_DELAYED) = 0; dr_set(X, full, BF_F_DONE, false);
		leaf_set_query(X, part, seed);
		cost_add_rss(X, full, part);
		if (X.ovf) return;
		const uint32_t na = AW(full + CA_NACT);
		AW(AW(full + CA_ACT) + na) = part; AW(full + CA_NACT) = na + 1u;
		dr_set_mincost(X, full, 0);
		cost_sort_actives(X, full);
		if (dr_found(X, full)) { dr_set(X, d, BF_F_FOUND, true); dr_set(X, full, BF_F_FOUND, false); }
	}
	if (dr_mincost(X, seed) > dr_mincost(X, d)) {
		uint32_t mc = dr_mincost(X, seed);
		if (!dr_done(X, full) && dr_mincost(X, full) < mc) mc = dr_mincost(X, full);
		dr_set_mincost(X, d, mc);
	}
}
/* seeded_advance from the return of cost_advance<1>(full) on (:3085-3100) */
BF_FN void seeded_post_full(BfLane& X, uint32_t d, uint32_t seed, uint32_t full, uint32_t old)
{
	if (dr_found(X, full)) { dr_set(X, d, BF_F_FOUND, true); dr_set(X, full, BF_F_FOUND, false); }
	if (dr_mincost(X, full) > old) {
		const uint32_t a = dr_mincost(X, full), b = dr_mincost(X, seed);
		dr_set_mincost(X, d, a < b ? a : b);
	}
}
/* cost_advance<0>(d) -- the aligner's own driver -- with the leaf it gets to advanced at one place (see the head of this
 * file): statement for statement cost_advance<0>, child_advance<0>, seeded_advance and cost_advance<1> up to their calls
 * (adv_pre), the call, then what follows it in each, innermost first (adv_post) */
struct BfAdvSt { uint32_t d, p, precost, leaf, seed, full, old, p2, precost1, after; };
enum { BF_AFTER_NONE = 0, BF_AFTER_SEED_BRANCH, BF_AFTER_FULL_BRANCH, BF_AFTER_FULL_CHILD };

/* the first halves, down to the leaf that is due (S.leaf, 0: none); false: the driver's advance ended here (a delayed range
 * handed out, nothing left to advance) and there are no second halves to go through */
BF_FNI bool adv_pre(BfLane& X, BfAdvSt& S, uint32_t d)
{
	S.d = d; S.leaf = 0; S.seed = 0; S.full = 0; S.old = 0; S.p2 = 0; S.precost1 = 0; S.after = BF_AFTER_NONE; S.p = 0; S.precost = 0;
	/* cost_advance<0>, first half */
	AW(d + CA_LAST) = 0;
	const uint32_t actSz = AW(d + CA_NACT);
	if (AW(d + CA_DELAYED)) {
		AW(d + CA_LAST) = AW(d + CA_DELAYED); AW(d + CA_DELAYED) = 0;
		dr_set(X, d, BF_F_FOUND, true);
		if (actSz > 0) { const uint32_t a0 = dr_mincost(X, AW(AW(d + CA_ACT))); if (a0 > dr_mincost(X, d)) dr_set_mincost(X, d, a0); }
		else dr_set(X, d, BF_F_DONE, true);
		return false;
	}
	if (cost_mate_eliminated<0>(X, d) || actSz == 0) { AW(d + CA_NACT) = 0; dr_set(X, d, BF_F_DONE, true); return false; }
	const uint32_t p = AW(AW(d + CA_ACT));
	S.p = p;
	S.precost = dr_mincost(X, p);
	if (!dr_found(X, p)) {
		if (dr_kind(X, p) != BF_SEEDED) S.leaf = p;          /* child_advance<0>: a leaf */
		else {
			/* seeded_advance(p), first half */
			const uint32_t seed = AW(p + SD_SEED), full = AW(p + SD_FULL);
			S.seed = seed; S.full = full;
			bool back = false;
			if (dr_done(X, seed) && dr_done(X, full) && !dr_found(X, seed) && !dr_found(X, full)) { dr_set(X, p, BF_F_DONE, true); back = true; }
			if (!back && dr_done(X, seed) && !dr_found(X, seed)) {
				dr_set_mincost(X, seed, 0xffffu);
				if (dr_mincost(X, full) > dr_mincost(X, p)) { dr_set_mincost(X, p, dr_mincost(X, full)); back = true; }
			}
			if (!back && dr_done(X, full) && !dr_found(X, full)) {
				dr_set_mincost(X, full, 0xffffu);
				if (dr_mincost(X, seed) > dr_mincost(X, p)) { dr_set_mincost(X, p, dr_mincost(X, seed)); back = true; }
			}
			if (!back) {
				if (dr_mincost(X, full) > dr_mincost(X, seed)) {
					S.after = BF_AFTER_SEED_BRANCH;
					if (!dr_found(X, seed)) S.leaf = seed;
				} else {
					S.after = BF_AFTER_FULL_BRANCH;
					S.old = dr_mincost(X, full);
					if (!dr_found(X, full)) {
						/* cost_advance<1>(full), first half (no mates to eliminate below the aligner's driver) */
						AW(full + CA_LAST) = 0;
						const uint32_t actSz1 = AW(full + CA_NACT);
						if (AW(full + CA_DELAYED)) {
							AW(full + CA_LAST) = AW(full + CA_DELAYED); AW(full + CA_DELAYED) = 0;
							dr_set(X, full, BF_F_FOUND, true);
							if (actSz1 > 0) { const uint32_t a0 = dr_mincost(X, AW(AW(full + CA_ACT))); if (a0 > dr_mincost(X, full)) dr_set_mincost(X, full, a0); }
							else dr_set(X, full, BF_F_DONE, true);
						} else if (actSz1 == 0) { AW(full + CA_NACT) = 0; dr_set(X, full, BF_F_DONE, true); }
						else {
							S.after = BF_AFTER_FULL_CHILD;
							S.p2 = AW(AW(full + CA_ACT));
							S.precost1 = dr_mincost(X, S.p2);
							if (!dr_found(X, S.p2)) S.leaf = S.p2;       /* child_advance<1>: always a leaf */
						}
					}
				}
			}
		}
	}
	return true;
}
/* the second halves, innermost first */
BF_FNI void adv_post(BfLane& X, BfAdvSt& S)
{
	uint32_t after = S.after;
	if (after == BF_AFTER_FULL_CHILD) { cost_advance_post<1>(X, S.full, S.p2, S.precost1); after = BF_AFTER_FULL_BRANCH; }
	if (after == BF_AFTER_FULL_BRANCH) seeded_post_full(X, S.p, S.seed, S.full, S.old);
	else if (after == BF_AFTER_SEED_BRANCH) seeded_post_seed(X, S.p, S.seed, S.full);
	cost_advance_post<0>(X, S.d, S.p, S.precost);
}
BF_FN void bf_advance_top(BfLane& X, uint32_t d)
{
	BfAdvSt S;
	if (!adv_pre(X, S, d)) return;
	if (S.leaf) leaf_advance(X, S.leaf);                  /* the one place */
	adv_post(X, S);
}

/* the static part of the tree (Unpaired*Factory::create()) */
BF_FN uint32_t bf_build_tree(BfLane& X)
{
	const BfProgram& P = *X.P;
	const uint32_t top = bf_alloc(X, BF_DRW);
	cost_init(X, top, P.strandFix, P.nnodes);
	for (uint32_t i = 0; i < P.nnodes && !X.ovf; i++) {
		const BfNode nd = P.nodes[i];
		const uint32_t d = bf_alloc(X, BF_DRW);
		if (nd.kind == BF_LEAF) leaf_init(X, d, nd.spec);
		else {
			const uint32_t gen = bf_alloc(X, BF_DRW), full = bf_alloc(X, BF_DRW);
			if (X.ovf) break;
			leaf_init(X, gen, nd.genSpec);
			cost_init(X, full, 0, 0);
			for (uint32_t k = 0; k < BF_DRW; k++) AW(d + k) = 0;
			AW(d + DR_KIND) = BF_SEEDED | ((uint32_t)nd.fw << 8) | ((uint32_t)P.specs[nd.spec].mate << 9);
			AW(d + DR_FLAGS) = BF_F_DONE;
			AW(d + SD_FULL) = full; AW(d + SD_SEED) = gen; AW(d + SD_FACT) = nd.spec;
		}
		cost_add_rss(X, top, d);
	}
	return top;
}

#if BF_HAVE_V1
/* one child of a cost-aware driver: a leaf, or a seeded driver with its seed generator and extender */
BF_FN uint32_t bf_make_node(BfLane& X, const BfNode nd)
{
	const BfProgram& P = *X.P;
	const uint32_t d = bf_alloc(X, BF_DRW);
	if (nd.kind == BF_LEAF) { if (!X.ovf) leaf_init(X, d, nd.spec); }
	else {
		const uint32_t gen = bf_alloc(X, BF_DRW), full = bf_alloc(X, BF_DRW);
		if (X.ovf) return d;
		leaf_init(X, gen, nd.genSpec);
		cost_init(X, full, 0, 0);
		for (uint32_t k = 0; k < BF_DRW; k++) AW(d + k) = 0;
		AW(d + DR_KIND) = BF_SEEDED | ((uint32_t)nd.fw << 8) | ((uint32_t)P.specs[nd.spec].mate << 9);
		AW(d + DR_FLAGS) = BF_F_DONE;
		AW(d + SD_FULL) = full; AW(d + SD_SEED) = gen; AW(d + SD_FACT) = nd.spec;
	}
	return d;
}

/* PairedBWAlignerV1's four drivers: tops[mate * 2 + (fw ? 0 : 1)] = driver1Fw, driver1Rc, driver2Fw, driver2Rc;
 * 0 = StubRangeSourceDriver (that strand of that mate is not searched: --nofw / --norc) */
BF_FN void bf_build_tree_v1(BfLane& X, uint32_t tops[4])
{
	const BfProgram& P = *X.P;
	uint32_t cnt[4] = {0, 0, 0, 0};
	for (uint32_t i = 0; i < P.nnodes; i++) {
		const BfSpec& sp = P.specs[P.nodes[i].spec];
		cnt[sp.mate * 2u + (sp.fw ? 0u : 1u)]++;
	}
	for (uint32_t b = 0; b < 4u; b++) {
		tops[b] = 0;
		if (cnt[b] && !X.ovf) { tops[b] = bf_alloc(X, BF_DRW); if (!X.ovf) cost_init(X, tops[b], P.strandFix, cnt[b]); }
	}
	for (uint32_t i = 0; i < P.nnodes && !X.ovf; i++) {
		const BfSpec& sp = P.specs[P.nodes[i].spec];
		const uint32_t d = bf_make_node(X, P.nodes[i]);
		if (X.ovf) break;
		cost_add_rss(X, tops[sp.mate * 2u + (sp.fw ? 0u : 1u)], d);
	}
}
#endif

#define BF_ADVANCE_TOP(X, d) do { BF_PT0(t_adv); bf_advance_top(X, d); BF_PADD(BP_ADV, t_adv); } while (0)

/* ---- RowChaser / RangeChaser (row_chaser.h:69-155, range_chaser.h:52-209; no range cache:
 * ebwt_search.cpp passes NULL caches) ------------------------------------------------------------- */
BF_FN void ch_row_set(BfLane& X, BfChase& c, bt_row row)
{
	const BtIndexDev& ix = X.ix[c.mirror];
	c.cRow = row;
	if (!BT_WIDE && ix.loc) {
		/* the index has its locus image (bt_rank.h): the row's offset from the dense suffix array, the walk RowChaser
		 * would have made (row_chaser.h:69-123) tallied from the table of walk lengths -- it ends at the '$' row, where no
		 * sample is read, exactly when it is as long as the offset */
		const uint32_t sa = BT_GP(const uint32_t, ix.loc)[(uint64_t)row * 4u];
		const uint32_t w = BT_GP(const uint16_t, ix.walk)[sa];     /* (32-bit rows only: the wide build has no locus image) */
		c.cOff = sa; c.cDone = 1; c.cJumps = w;
		X.c_chase += w;
		if (sa != w) X.c_offs++;
		return;
	}
	if (row == ix.zOff) { c.cOff = 0; c.cDone = 1; return; }
	if ((row & ix.offMask) == row) { c.cOff = BT_GP(const bt_row, ix.offs)[row >> ix.offRate]; c.cDone = 1; X.c_offs++; return; }
	c.cDone = 0; c.cJumps = 0; c.cOff = BT_OFF_MASK;
}
BF_FN void ch_row_off(BfLane& X, BfChase& c)
{
	uint32_t tidx = BF_NONE32, toff = BF_NONE32;
	if (!bt_joined_to_text(X.ix[c.mirror], c.qlen, c.cOff, &tidx, &toff, &X.c_rst)) tidx = BF_NONE32;
	c.tidx = tidx; c.toff = toff;
}
BF_FN void ch_set_row(BfLane& X, BfChase& c, bt_row row)
{
	c.row = row;
	for (;;) {
		ch_row_set(X, c, c.row);
		if (!c.cDone) break;
		ch_row_off(X, c);
		if (c.tidx != BF_NONE32) return;
		c.row++;
		if (c.row == c.bot) c.row = c.top;
		if (c.row == c.irow) { c.done = 1; return; }
	}
}
BF_FN void ch_set_top_bot(BfLane& X, BfChase& c, bt_row top, bt_row bot, uint32_t mirror, uint32_t qlen)
{
	BF_PT0(t_chase);
	c.mirror = mirror; c.qlen = qlen; c.top = top; c.bot = bot;
	c.irow = top + (bf_rnd(X.alRnd) % (bot - top));
	c.done = 0; c.tidx = BF_NONE32;
	ch_set_row(X, c, c.irow);
	BF_PADD(BP_CHASE, t_chase);
}
/* one row of the walk resolved, or one LF step of it: RangeChaser::advance (range_chaser.h:150-209) a piece at a time --
 * ch_advance below is this in a loop, the wavefront automaton takes a piece per round */
BF_FNI void ch_advance_piece(BfLane& X, BfChase& c)
{
	c.tidx = BF_NONE32;
	if (c.cDone) {
		c.row++;
		if (c.row == c.bot) c.row = c.top;
		if (c.row == c.irow) c.done = 1;
		else ch_set_row(X, c, c.row);
	} else {
		/* RowChaser::advance: one LF step */
		const BtIndexDev& ix = X.ix[c.mirror];
		bt_row lf[4]; uint32_t L;
		bt_rank4(ix, c.cRow, lf, &L);
		c.cRow = lf[L];
		c.cJumps++; X.c_chase++;
		if (c.cRow == ix.zOff) { c.cOff = c.cJumps; c.cDone = 1; }
		else if ((c.cRow & ix.offMask) == c.cRow) { c.cOff = BT_GP(const bt_row, ix.offs)[c.cRow >> ix.offRate] + c.cJumps; c.cDone = 1; X.c_offs++; }
		if (c.cDone) ch_row_off(X, c);
	}
}
BF_FN void ch_advance(BfLane& X, BfChase& c)
{
	BF_PT0(t_chase);
	const bool walking = !c.cDone;
	do ch_advance_piece(X, c); while (walking && !c.cDone);
	BF_PADD(BP_CHASE, t_chase);
}

/* ---- sink + hit record -------------------------------------------------------------------------
 * UnpairedAlignerV2::report (aligner.h:467-497) / PairedBWAlignerV2::report (:1720-1788),
 * EbwtSearchParams::reportHit (ebwt.h:1288-1405), NGood / NBestFirstStrat / All sinks
 * (hit.h:969-985, 1070-1129, 1201-1209).  true = stop.
 * getmm(i, &m, &refc): the i-th mismatch as (offset in the searched query string, reference base);
 * `flip` says that string runs 3'->5' (ebwtFw != fw). */
template <class F>
BF_FN bool bf_emit_hit(BfLane& X, const BtBatchDev& B, uint32_t fw, bool flip, uint32_t alen, uint32_t cost, uint32_t oms,
                       uint32_t mate, uint32_t nmm, uint32_t tidx, uint32_t toff, F getmm)
{
	const BfProgram& P = *X.P;
	const uint32_t stratum = cost >> 14;
	X.nhits++;
	if (P.sinkStrata && stratum < X.bestStratum) X.bestStratum = stratum;
	if (X.nhits > P.sinkMax) return true;
	if (X.stored < B.hit_cap) {
		BtHitRec h;
		h.tidx = tidx; h.toff = toff; h.oms = oms;
		h.cost = (uint16_t)cost; h.stratum = (uint8_t)stratum; h.fw = (uint8_t)fw; h.pad[0] = (uint8_t)mate; h.pad[1] = 0;
		h.nmm = (uint16_t)nmm; h.mm_off = 0;
		if (nmm > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
			const uint32_t off = atomicAdd(B.mm_pool_used, nmm);
#else
			const uint32_t off = *B.mm_pool_used; *B.mm_pool_used += nmm;
#endif
			if (off + nmm <= B.mm_pool_cap) {
				h.mm_off = off;
				auto mm = BT_GP(uint16_t, B.mm_pool + off);
				/* up to 16 mismatches are put in order in the lane and stored once: the pool is never read (sorting in place
				 * waits for its own stores to come back, and reads a line other lanes' lists share) */
				uint16_t ee[16];
				if (nmm <= 16u) {
					for (uint32_t i = 0; i < nmm; i++) {
						uint32_t m, refc;
						getmm(i, m, refc);
						const uint32_t pos = flip ? alen - m - 1u : m;
						const uint16_t e16 = (uint16_t)(pos | (refc << 12));
						int j = (int)i - 1;
						while (j >= 0 && (ee[j] & 0x3ffu) > (e16 & 0x3ffu)) { ee[j + 1] = ee[j]; j--; }
						ee[j + 1] = e16;
					}
					for (uint32_t i = 0; i < nmm; i++) mm[i] = ee[i];
				} else
				for (uint32_t i = 0; i < nmm; i++) {
					uint32_t m, refc;
					getmm(i, m, refc);
					const uint32_t pos = flip ? alen - m - 1u : m;
					const uint16_t e16 = (uint16_t)(pos | (refc << 12));
					int j = (int)i - 1;                              /* Hit::mms is a bitset: ordered by position */
					while (j >= 0 && (mm[j] & 0x3ffu) > (e16 & 0x3ffu)) { mm[j + 1] = mm[j]; j--; }
					mm[j + 1] = e16;
				}
			} else { h.nmm = 0; X.status |= BT_STF_MMPOOL; }
		}
		uint32_t hw[6];
		__builtin_memcpy(hw, &h, 24);
		uint32_t* dst = (uint32_t*)(B.hits + ((uint64_t)X.rd * B.hit_cap + X.stored));
		BtU4 q; q.x = hw[0]; q.y = hw[1]; q.z = hw[2]; q.w = hw[3];
		bt_st4(dst, q);
		BT_GP(uint32_t, dst)[4] = hw[4]; BT_GP(uint32_t, dst)[5] = hw[5];
		X.stored++;
	} else if (X.stored < P.sinkN) X.status |= BT_STF_HITCAP;
	if (P.sinkAll && !P.sinkStrata) return false;
	return X.nhits == P.sinkN && (P.sinkMax == 0xffffffffu || P.sinkMax < P.sinkN);
}

/* a hit from a leaf's current range: its edits are the found branch's chain plus the seed range's */
BF_FN bool bf_report_leaf_body(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff, uint32_t mate, uint32_t oms, bool ebwtFw);
BF_FN bool bf_report_leaf(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff, uint32_t mate,
                          uint32_t oms, bool ebwtFw)
{
	BF_PT0(t_report);
	const bool r = bf_report_leaf_body(X, B, leaf, tidx, toff, mate, oms, ebwtFw);
	BF_PADD(BP_REPORT, t_report);
	return r;
}
BF_FN bool bf_report_leaf_body(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff, uint32_t mate,
                               uint32_t oms, bool ebwtFw)
{
	const BfSpec& sp = leaf_spec(X, leaf);
	const uint32_t cost = AW(leaf + LF_CURCOST) & 0xffffu, nmm = AW(leaf + LF_CURCOST) >> 16;
	const uint32_t qlen = AW(leaf + LF_QLEN) & 0xffffu;
	const uint32_t sn = (AW(leaf + LF_RSFLAGS) & 8u) ? (AW(leaf + LF_SEED) >> 16) : 0u;
	uint32_t b = AW(leaf + LF_CURBR);
	return bf_emit_hit(X, B, sp.fw, ebwtFw != (sp.fw != 0), X.R[sp.mate].len, cost, oms, mate, nmm, tidx, toff,
		[&](uint32_t i, uint32_t& m, uint32_t& refc) {
			if (i < nmm - sn) { const uint32_t e = AW(b + BR_EDIT); m = qlen - (e & 0x3ffu) - 1u; refc = (e >> 10) & 3u; b = AW(b + BR_PARENT); }
			else { const uint32_t sd = AW(leaf + LF_SEEDMM0 + (i - (nmm - sn))); m = qlen - (sd & 0xffffu) - 1u; refc = sd >> 16; }
		});
}
BF_INL bool bf_irrelevant(const BfLane& X, uint32_t cost)       /* NBestFirstStrat::irrelevantCost (hit.h:1121-1127) */
{
	return X.P->sinkStrata && X.nhits && (cost >> 14) > X.bestStratum;
}

/* does the read hold an N (code 4)?  Its row in 16-byte pieces (rows are 16-byte aligned and padded with 4s past the
 * read's end, which are masked off) */
BF_FN uint32_t bf_has_n(const BfRead& R)
{
	uint32_t any = 0;
	for (uint32_t base = 0; base < R.len; base += 16u) {
		const BtU4 v = bt_ld4((const void*)(R.seq + base));
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		for (uint32_t k = 0; k < 4u; k++) {
			const uint32_t at = base + 4u * k;
			if (at >= R.len) break;
			const uint32_t t = w[k] ^ 0x04040404u;
			uint32_t z = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;      /* 0x80 per byte that is 4 */
			const uint32_t left = R.len - at;
			if (left < 4u) z &= (1u << (8u * left)) - 1u;
			any |= z;
		}
	}
	return any != 0 ? 1u : 0u;
}

BF_FN void bf_read_begin(BfLane& X, const BtBatchDev& B, uint32_t rd)
{
	X.rd = rd;
	X.R[0].len = BT_GP(const uint16_t, B.len)[rd];
	X.R[0].seed = BT_GP(const uint32_t, B.seed)[rd];
	X.R[0].seq = (BF_G const uint8_t*)(B.seq + (uint64_t)rd * B.stride);
	X.R[0].qual = (BF_G const uint8_t*)(B.qual + (uint64_t)rd * B.stride);
	X.R[1] = X.R[0];
	if (B.seq2) {
		X.R[1].len = BT_GP(const uint16_t, B.len2)[rd];
		X.R[1].seed = BT_GP(const uint32_t, B.seed2)[rd];
		X.R[1].seq = (BF_G const uint8_t*)(B.seq2 + (uint64_t)rd * B.stride2);
		X.R[1].qual = (BF_G const uint8_t*)(B.qual2 + (uint64_t)rd * B.stride2);
	}
	X.top = BF_RESERVED; X.ovf = 0; X.growing = 0;
	X.nhits = 0; X.stored = 0; X.bestStratum = 999; X.status = 0;
	X.alRnd = X.R[0].seed;                                      /* Aligner::rand_.init(bufa_->seed) */
	X.btCnt = (int32_t)X.P->maxBts;
	X.hasN = bf_has_n(X.R[0]) | (B.seq2 ? bf_has_n(X.R[1]) << 1 : 0u);
}
BF_FN void bf_read_end(BfLane& X, const BtBatchDev& B, uint32_t mult)
{
	BF_PT0(t_end);
	if (X.ovf) X.status |= BT_STF_OVERFLOW;
	/* NBestFirstStrat::finishReadImpl (hit.h:1098-1110): every buffered hit's oms = #buffered / mult - 1 */
	if (X.P->sinkStrata) {
		for (uint32_t k = 0; k < X.stored; k++)
			BT_GP(uint32_t, (uint32_t*)(B.hits + ((uint64_t)X.rd * B.hit_cap + k)))[2] = X.stored / mult - 1u;
	}
	BT_GP(uint32_t, B.n_hits)[X.rd] = X.nhits;
	BT_GP(uint8_t, B.status)[X.rd] = (uint8_t)X.status;
	BF_PADD(BP_END, t_end);
}
BF_INL void bf_chase_init(BfChase& ch)
{
	ch.mirror = 0; ch.qlen = 0; ch.top = ch.bot = ch.irow = ch.row = 0; ch.tidx = BF_NONE32; ch.toff = 0;
	ch.done = 0; ch.cDone = 1; ch.cRow = 0; ch.cJumps = 0; ch.cOff = 0;
}

/* ---- one read: UnpairedAlignerV2::setQuery + advance() until done (aligner.h:434-567) ---------- */
BF_FN void bf_run_read(BfLane& X, const BtBatchDev& B, uint32_t rd)
{
	BF_PT0(t_begin);
	bf_read_begin(X, B, rd);
	if (X.R[0].len < 4u) {
		X.status |= BT_STF_SKIPPED;
	} else {
		const uint32_t drv = bf_build_tree(X);
		BfChase ch;
		bf_chase_init(ch);
		bool done = true, chase = false;
		if (!X.ovf) { cost_set_query<0>(X, drv); done = dr_done(X, drv); }
		BF_PADD(BP_BEGIN, t_begin);
		while (!done && !X.ovf) {
			if (chase) {
				if (ch.tidx == BF_NONE32 && !ch.done) { ch_advance(X, ch); continue; }
				if (ch.tidx != BF_NONE32) {
					const uint32_t leaf = AW(drv + CA_LAST);
					done = bf_report_leaf(X, B, leaf, ch.tidx, ch.toff, 0, (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP) - 1u),
					                      !leaf_spec(X, leaf).mirror);
					ch.tidx = BF_NONE32;
				} else {
					chase = false;
					dr_set(X, drv, BF_F_FOUND, false);
					done = dr_done(X, drv);
				}
			}
			if (!done && !chase) {
				if (dr_found(X, drv)) {
					const uint32_t leaf = AW(drv + CA_LAST);
					const uint32_t cost = AW(leaf + LF_CURCOST) & 0xffffu;
					ch_set_top_bot(X, ch, AR(leaf + LF_CURTOP), AR(leaf + LF_CURBOT), leaf_spec(X, leaf).mirror, X.R[0].len);
					if (ch.tidx != BF_NONE32) {
						done = bf_report_leaf(X, B, leaf, ch.tidx, ch.toff, 0, (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP) - 1u),
						                      !leaf_spec(X, leaf).mirror);
						ch.tidx = BF_NONE32;
					}
					if (!ch.done && !bf_irrelevant(X, cost)) chase = true;
					else dr_set(X, drv, BF_F_FOUND, false);
				} else {
					done = bf_irrelevant(X, dr_mincost(X, drv));
					if (!done) BF_ADVANCE_TOP(X, drv);
				}
				if (dr_done(X, drv) && !dr_found(X, drv) && !chase) done = true;
			}
		}
	}
	bf_read_end(X, B, 1u);
}

/* ---- paired-end -------------------------------------------------------------------------------- */
BF_INL uint32_t ref_base(const BtRefDev& rf, uint64_t p)        /* 0..3, or 4 for N / gap (BitPairReference::getStretch) */
{
	if ((BT_GP(const uint32_t, rf.nmask)[p >> 5] >> (p & 31u)) & 1u) return 4u;
	return (BT_GP(const uint32_t, rf.bits)[p >> 4] >> (2u * (uint32_t)(p & 15u))) & 3u;
}

/* RefAligner::find with numToFind = 1 (ref_aligner.h:63-101; specification = the naiveFind of the
 * concrete aligners, :182-262, 513-601, 2561-2752): candidate leftmost positions radiate out from
 * the middle of [begin, end - qlen]; the first one that touches no reference N, keeps the seed's
 * (or, -v, the read's) mismatches within refMms and -- seeded -- the summed penalties within
 * refQualMax, and has not been reported yet for this pair orientation (TSetPairs), is it.
 * Its mismatches go to AW(mmBuf ..) as m | refc<<16; returns true and sets the out-parameters. */
BF_FN bool bf_ref_find_one(BfLane& X, uint32_t tidx, const BfRead& M, uint32_t fw, uint32_t begin, uint32_t end,
                           uint32_t pairsList /* arena words: [0] n, [1] cap, [2] array */, uint32_t aoff, uint32_t mmBuf,
                           uint32_t* result, uint32_t* nmmOut, uint32_t* stratumOut)
{
	const BfProgram& P = *X.P;
	const BtRefDev& rf = *X.ref;
	const uint32_t qlen = M.len;
	const uint32_t slen = P.refSeeded ? (qlen < P.refSeedLen ? qlen : P.refSeedLen) : qlen;
	const uint64_t base = BT_GP(const uint64_t, rf.start)[tidx];
	const uint32_t lim = end - qlen - begin, halfway = begin + (lim >> 1);
	/* Packed pre-filter (reads of up to 64 bases): the mate as 2-bit codes in two 64-bit words, the reference under each
	 * candidate position cut out of three 64-bit pieces of the 2-bit reference that stay in registers while the
	 * candidates walk away from the middle (one set for the positions below it, one for those above: a new piece every
	 * 32 candidates) -- one XOR + popcount says whether the seed's mismatches can be within refMms, one AND whether the
	 * span touches an N.  Only candidates that pass are then looked at base by base below, which decides exactly as
	 * before (the reference's naive scan compares base by base throughout). */
	const bool packed = qlen <= 64u;
	uint64_t rq[2] = {0, 0}, rn[2] = {0, 0}, sm[2] = {0, 0}, lm[2] = {0, 0};
	uint64_t W[2][3] = {{0, 0, 0}, {0, 0, 0}}, N[2][2] = {{0, 0}, {0, 0}}, K[2] = {~0ull, ~0ull};
	/* the mate's bases (and, for -n, qualities) as they are stored, in registers: 16-byte pieces of its rows instead of a
	 * fetch per character here and in the base-by-base decision below */
	uint32_t mw[16], qw[16];
	for (uint32_t k = 0; k < 16u; k++) { mw[k] = 0; qw[k] = 0; }
	if (packed) {
		for (uint32_t c = 0; c * 16u < qlen; c++) {
			const BtU4 v = bt_ld4((const void*)(M.seq + 16u * c));
			mw[4u * c] = v.x; mw[4u * c + 1u] = v.y; mw[4u * c + 2u] = v.z; mw[4u * c + 3u] = v.w;
			if (P.refSeeded) {
				const BtU4 u = bt_ld4((const void*)(M.qual + 16u * c));
				qw[4u * c] = u.x; qw[4u * c + 1u] = u.y; qw[4u * c + 2u] = u.z; qw[4u * c + 3u] = u.w;
			}
		}
	}
	auto mbase = [&](uint32_t j) -> uint32_t {             /* bf_base(M, fw, 1, j) */
		const uint32_t idx = fw ? j : qlen - 1u - j;
		const uint32_t c = (mw[idx >> 2] >> (8u * (idx & 3u))) & 0xffu;
		return (!fw && c < 4u) ? (c ^ 3u) : c;
	};
	auto mqual = [&](uint32_t j) -> uint32_t {             /* bf_qualc(M, fw, 1, j) */
		const uint32_t idx = fw ? j : qlen - 1u - j;
		return (qw[idx >> 2] >> (8u * (idx & 3u))) & 0xffu;
	};
	if (packed) {
		for (uint32_t j = 0; j < qlen; j++) {
			const uint32_t q = mbase(j);
			const uint64_t bit = 1ull << (2u * (j & 31u));
			if (q < 4u) rq[j >> 5] |= (uint64_t)q << (2u * (j & 31u)); else rn[j >> 5] |= bit;
			lm[j >> 5] |= bit;
			if (fw ? (j < slen) : (j >= qlen - slen)) sm[j >> 5] |= bit;
		}
	}
	const uint64_t nspan = qlen >= 64u ? ~0ull : ((1ull << qlen) - 1ull);
	uint64_t pm0 = 0, pm1 = 0, pa0 = 0, pa1 = 0;           /* the candidate's mismatch masks and reference words */
	bool hi = false;
	for (uint32_t i = 1; i <= lim + 1u; i++) {
		const uint32_t ri = hi ? halfway + (i >> 1) : halfway - (i >> 1);
		const uint32_t side = hi ? 1u : 0u;
		hi = !hi;
		if (packed) {
			const uint64_t p = base + ri;
			if ((p >> 5) != K[side]) {
				K[side] = p >> 5;
				BF_G const uint64_t* b64 = (BF_G const uint64_t*)rf.bits; BF_G const uint64_t* n64 = (BF_G const uint64_t*)rf.nmask;
				W[side][0] = b64[K[side]]; W[side][1] = b64[K[side] + 1u]; W[side][2] = b64[K[side] + 2u];
				N[side][0] = n64[p >> 6]; N[side][1] = n64[(p >> 6) + 1u];
			}
			const uint32_t t = (uint32_t)(p & 63u), s2 = 2u * (uint32_t)(p & 31u);
			const uint64_t nn = t ? (N[side][0] >> t) | (N[side][1] << (64u - t)) : N[side][0];
			if (nn & nspan) continue;                                      /* a reference N / gap under the mate */
			const uint64_t a0 = s2 ? (W[side][0] >> s2) | (W[side][1] << (64u - s2)) : W[side][0];
			const uint64_t a1 = s2 ? (W[side][1] >> s2) | (W[side][2] << (64u - s2)) : W[side][1];
			const uint64_t x0 = a0 ^ rq[0], x1 = a1 ^ rq[1];
			const uint64_t m0 = (((x0 | (x0 >> 1)) & 0x5555555555555555ull) | rn[0]) & lm[0];
			const uint64_t m1 = (((x1 | (x1 >> 1)) & 0x5555555555555555ull) | rn[1]) & lm[1];
			if ((uint32_t)__builtin_popcountll(m0 & sm[0]) + (uint32_t)__builtin_popcountll(m1 & sm[1]) > P.refMms) continue;
			pm0 = m0; pm1 = m1; pa0 = a0; pa1 = a1;
		}
		bool match = true;
		uint32_t mms = 0, seedMms = 0, ham = 0;
		if (packed) {
			/* the same decision from the registers: the mismatching offsets are the set bits of the two masks, in ascending
			 * order; the reference has no N under the mate (checked above) */
			for (uint32_t w = 0; w < 2u && match; w++) {
				uint64_t bits = w ? pm1 : pm0;
				const uint64_t aw = w ? pa1 : pa0;
				while (bits) {
					const uint32_t b = (uint32_t)__builtin_ctzll(bits);
					bits &= bits - 1ull;
					const uint32_t j = 32u * w + (b >> 1), rc = (uint32_t)(aw >> b) & 3u;
					const bool inSeed = fw ? (j < slen) : (j >= qlen - slen);
					if (inSeed && ++seedMms > P.refMms) { match = false; break; }
					if (P.refSeeded) {
						ham += bt_mm_penalty(P.maq, bf_phred(mqual(j)));
						if (ham > P.refQualMax) { match = false; break; }
					}
					AW(mmBuf + mms) = j | (rc << 16); mms++;
				}
			}
		} else
		for (uint32_t j = 0; j < qlen; j++) {
			const uint32_t rc = ref_base(rf, base + ri + j);
			if (rc & 4u) { match = false; break; }
			const uint32_t q = bf_base(M, fw, 1u, j);               /* patFw / patRc */
			if (q != rc) {
				const bool inSeed = fw ? (j < slen) : (j >= qlen - slen);
				if (inSeed && ++seedMms > P.refMms) { match = false; break; }
				if (P.refSeeded) {
					ham += bt_mm_penalty(P.maq, bf_phred(bf_qualc(M, fw, 1u, j)));
					if (ham > P.refQualMax) { match = false; break; }
				}
				AW(mmBuf + mms) = j | (rc << 16); mms++;
			}
		}
		if (!match) continue;
		/* TSetPairs: (upstream, downstream) coordinates already reported for this orientation */
		const uint32_t first = ri < aoff ? ri : aoff, second = ri < aoff ? aoff : ri;
		uint32_t n = AW(pairsList), cap = AW(pairsList + 1u), arr = AW(pairsList + 2u);
		bool dup = false;
		for (uint32_t k = 0; k < n && !dup; k++) dup = AW(arr + 3u * k) == tidx && AW(arr + 3u * k + 1u) == first && AW(arr + 3u * k + 2u) == second;
		if (dup) continue;
		if (n == cap) {
			const uint32_t ncap = cap ? cap * 2u : 8u;
			const uint32_t na = bf_alloc(X, 3u * ncap);
			if (X.ovf) return false;
			for (uint32_t k = 0; k < 3u * n; k++) AW(na + k) = AW(arr + k);
			arr = na; AW(pairsList + 1u) = ncap; AW(pairsList + 2u) = arr;
		}
		AW(arr + 3u * n) = tidx; AW(arr + 3u * n + 1u) = first; AW(arr + 3u * n + 2u) = second;
		AW(pairsList) = n + 1u;
		*result = ri; *nmmOut = mms; *stratumOut = seedMms;
		return true;
	}
	return false;
}

/* PairedBWAlignerV2::resolveOutstandingInRef + report (aligner.h:1883-1997, 1720-1788): the anchor
 * mate's range `leaf` resolved to (tidx, toff); look for the other mate in the window the insert
 * constraints allow and report the pair, upstream mate first.  true = the sink says stop. */
BF_FN bool bf_resolve_in_ref_body(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff, uint32_t pairsFw, uint32_t pairsRc, uint32_t mmBuf);
BF_FN bool bf_resolve_in_ref(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff,
                             uint32_t pairsFw, uint32_t pairsRc, uint32_t mmBuf)
{
	BF_PT0(t_ref);
	const bool r = bf_resolve_in_ref_body(X, B, leaf, tidx, toff, pairsFw, pairsRc, mmBuf);
	BF_PADD(BP_REF, t_ref);
	return r;
}
BF_FN bool bf_resolve_in_ref_body(BfLane& X, const BtBatchDev& B, uint32_t leaf, uint32_t tidx, uint32_t toff,
                                  uint32_t pairsFw, uint32_t pairsRc, uint32_t mmBuf)
{
	const BfProgram& P = *X.P;
	const BfSpec& sp = leaf_spec(X, leaf);
	const bool amate1 = sp.mate == 0, afw = sp.fw != 0;
	const bool pairFw = amate1 ? (afw == (P.mate1Fw != 0)) : (afw == (P.mate2Fw != 0));
	const bool matchRight = pairFw ? amate1 : !amate1;
	bool fw = amate1 ? (P.mate2Fw != 0) : (P.mate1Fw != 0);
	if (!pairFw) fw = !fw;
	const BfRead& M = X.R[amate1 ? 1 : 0];                       /* the outstanding mate */
	const uint32_t qlen = M.len, alen = X.R[sp.mate].len;
	const uint32_t minins = P.minIns, maxins = P.maxIns;
	if (maxins <= (qlen > alen ? qlen : alen)) return false;
	const uint32_t approx = BT_GP(const uint32_t, X.ref->approxLen)[tidx];
	uint32_t begin, end;
	if (matchRight) {
		const uint32_t insDiff = maxins - minins;
		end = toff + maxins;
		begin = toff + (P.allowContain ? 0u : 1u);
		if (!P.allowContain && qlen < alen) begin += alen - qlen;
		if (end > insDiff + qlen) { const uint32_t b2 = end - insDiff - qlen; if (b2 > begin) begin = b2; }
		if (approx < end) end = approx;
		if (approx < begin) begin = approx;
	} else {
		begin = (toff + alen < maxins) ? 0u : toff + alen - maxins;
		const uint32_t mi = alen < qlen ? alen : qlen;
		if (P.allowContain) end = toff + alen - (BF_IS_V1(P) ? 0u : 1u);       /* aligner.h:1046 (V1) / :1958 (V2) */
		else {
			end = toff + mi - 1u;
			const uint32_t e2 = toff + alen - minins + qlen - 1u;
			if (e2 < end) end = e2;
			if (toff + alen + qlen < minins + 1u) end = 0;
		}
	}
	if (end - begin < qlen || end < begin) return false;
	uint32_t result = 0, nmm = 0, stratum = 0;
	if (!bf_ref_find_one(X, tidx, M, fw ? 1u : 0u, begin, end, pairFw ? pairsFw : pairsRc, toff, mmBuf, &result, &nmm, &stratum)) return false;
	const uint32_t oms = (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP) - 1u);     /* both mates carry the anchor's range */
	const uint32_t omate = amate1 ? 1u : 0u;
	/* the found mate: fw-index coordinates (ebwtFw = true), cost = stratum << 14 (aligner.h:1973) */
	auto emit_found = [&](uint32_t mateNo) {
		return bf_emit_hit(X, B, fw ? 1u : 0u, !fw, qlen, stratum << 14, oms, mateNo, nmm, tidx, result,
			[&](uint32_t i, uint32_t& m, uint32_t& refc) { const uint32_t w = AW(mmBuf + i); m = w & 0xffffu; refc = w >> 16; });
	};
	(void)omate;
	const uint32_t mateL = pairFw ? 1u : 2u, mateR = pairFw ? 2u : 1u;
	if (matchRight) {
		if (bf_report_leaf(X, B, leaf, tidx, toff, mateL, oms, !sp.mirror)) return true;
		return emit_found(mateR);
	}
	if (emit_found(mateL)) return true;
	return bf_report_leaf(X, B, leaf, tidx, toff, mateR, oms, !sp.mirror);
}

/* PairedBWAlignerV2::setQuery + advance() until done (aligner.h:1571-1701), reportSe off */
BF_FN void bf_run_pair(BfLane& X, const BtBatchDev& B, uint32_t rd)
{
	BF_PT0(t_begin);
	bf_read_begin(X, B, rd);
	if (X.R[0].len < 4u || X.R[1].len < 4u) {
		X.status |= BT_STF_SKIPPED;
	} else {
		const uint32_t drv = bf_build_tree(X);
		const uint32_t maxLen = X.R[0].len > X.R[1].len ? X.R[0].len : X.R[1].len;
		const uint32_t pairsFw = bf_alloc(X, 3), pairsRc = bf_alloc(X, 3), mmBuf = bf_alloc(X, maxLen);
		BfChase ch;
		bf_chase_init(ch);
		bool done = true, chase = false;
		uint32_t attempts = 0;
		if (!X.ovf) {
			AW(pairsFw) = AW(pairsFw + 1u) = AW(pairsFw + 2u) = 0; AW(pairsRc) = AW(pairsRc + 1u) = AW(pairsRc + 2u) = 0;
			cost_set_query<0>(X, drv);
			done = false;
		}
		BF_PADD(BP_BEGIN, t_begin);
		while (!done && !X.ovf) {
			if (chase) {
				if (ch.tidx == BF_NONE32 && !ch.done) { ch_advance(X, ch); continue; }
				if (ch.tidx != BF_NONE32) {
					/* resolveOutstanding (aligner.h:1849-1871) */
					const bool ret = bf_resolve_in_ref(X, B, AW(drv + CA_LAST), ch.tidx, ch.toff, pairsFw, pairsRc, mmBuf);
					if (++attempts > X.P->pairTries || ret) done = true;
					ch.tidx = BF_NONE32;
				} else {
					chase = false;
					done = dr_done(X, drv);
				}
			}
			if (!done && !chase) {
				if (!dr_done(X, drv)) {
					done = bf_irrelevant(X, dr_mincost(X, drv));
					if (!done) BF_ADVANCE_TOP(X, drv);
					if (dr_found(X, drv)) {
						chase = true;
						dr_set(X, drv, BF_F_FOUND, false);
						const uint32_t leaf = AW(drv + CA_LAST);
						const BfSpec& sp = leaf_spec(X, leaf);
						ch_set_top_bot(X, ch, AR(leaf + LF_CURTOP), AR(leaf + LF_CURBOT), sp.mirror, X.R[sp.mate].len);
					}
				} else done = true;
			}
		}
	}
	bf_read_end(X, B, 2u);
}

#if BF_HAVE_V1
/* PairedBWAlignerV1::setQuery + advance() until done (aligner.h:726-847, advanceOrientation :1091-1320) with
 * dontReconcileMates (the default, ebwt_search.cpp:219): first the pairing in which mate 1 lies on its own strand
 * (L = mate 1, R = mate 2), then the other (L = mate 2, R = mate 1); every offset found for a range of one mate goes
 * to the reference scan for the other (bf_resolve_in_ref). */
struct BfV1Orient { uint32_t drL, drR; bool chaseL, chaseR, delayedL, delayedR; uint32_t szL, szR; };

BF_FN void bf_run_pair_v1(BfLane& X, const BtBatchDev& B, uint32_t rd)
{
	bf_read_begin(X, B, rd);
	if (X.R[0].len < 4u || X.R[1].len < 4u) {
		X.status |= BT_STF_SKIPPED;
	} else {
		const BfProgram& P = *X.P;
		uint32_t tops[4];
		bf_build_tree_v1(X, tops);
		const uint32_t maxLen = X.R[0].len > X.R[1].len ? X.R[0].len : X.R[1].len;
		const uint32_t pairsFw = bf_alloc(X, 3), pairsRc = bf_alloc(X, 3), mmBuf = bf_alloc(X, maxLen);
		BfChase ch;
		bf_chase_init(ch);
		bool done = true;
		if (!X.ovf) {
			AW(pairsFw) = AW(pairsFw + 1u) = AW(pairsFw + 2u) = 0; AW(pairsRc) = AW(pairsRc + 1u) = AW(pairsRc + 2u) = 0;
			for (uint32_t b = 0; b < 4u && !X.ovf; b++) if (tops[b]) cost_set_query<0>(X, tops[b]);
			done = false;
		}
		const bool fw1 = P.mate1Fw != 0, fw2 = P.mate2Fw != 0;
		BfV1Orient O[2];
		O[0].drL = fw1 ? tops[0] : tops[1]; O[0].drR = fw2 ? tops[2] : tops[3];      /* aligner.h:670-682 */
		O[1].drL = fw2 ? tops[3] : tops[2]; O[1].drR = fw1 ? tops[1] : tops[0];      /* aligner.h:684-696 */
		for (int k = 0; k < 2; k++) { O[k].chaseL = O[k].chaseR = O[k].delayedL = O[k].delayedR = false; O[k].szL = O[k].szR = 0; }
		const uint32_t qlen1 = X.R[0].len, qlen2 = X.R[1].len;
		const uint32_t symCeil = P.sinkMax == 0xffffffffu ? 0xffffffffu : P.sinkMax / 2u;   /* -m ("mhits, // for symCeiling") */
		uint32_t attempts = 0, o = 0;
		bool doneFw = false, doneFwFirst = true;
#define V1_DONE(d)  ((d) == 0u || dr_done(X, (d)))
		auto chase_range_of = [&](uint32_t top, uint32_t qlen) {
			const uint32_t leaf = AW(top + CA_LAST);
			ch_set_top_bot(X, ch, AR(leaf + LF_CURTOP), AR(leaf + LF_CURBOT), leaf_spec(X, leaf).mirror, qlen);
		};
		while (!done && !X.ovf) {
			if (doneFw && doneFwFirst) { o = 1; doneFwFirst = false; attempts = 0; }
			BfV1Orient& Q = O[o];
			if ((Q.chaseL || Q.chaseR) && ch.tidx == BF_NONE32 && !ch.done) { ch_advance(X, ch); continue; }
			bool& donePair = (o == 0) ? doneFw : done;
			bool returned = false;
			if (Q.chaseL || Q.chaseR) {
				const bool sideL = Q.chaseL;
				const uint32_t drMe = sideL ? Q.drL : Q.drR, drOther = sideL ? Q.drR : Q.drL;
				bool& chaseMe = sideL ? Q.chaseL : Q.chaseR;
				bool& chaseOther = sideL ? Q.chaseR : Q.chaseL;
				bool& delayedOther = sideL ? Q.delayedR : Q.delayedL;
				if (ch.tidx != BF_NONE32) {
					if (!done) {
						done = bf_resolve_in_ref(X, B, AW(drMe + CA_LAST), ch.tidx, ch.toff, pairsFw, pairsRc, mmBuf);
						if (++attempts > P.pairTries) { donePair = true; returned = true; }
					}
					if (!returned) ch.tidx = BF_NONE32;                       /* rchase_->reset() */
				} else {
					chaseMe = false;
					dr_set(X, drMe, BF_F_FOUND, false);
					if (delayedOther) {
						chase_range_of(drOther, sideL ? (doneFw ? qlen1 : qlen2) : (doneFw ? qlen2 : qlen1));
						chaseOther = true; delayedOther = false;
					}
				}
			}
			if (returned) continue;
			if (!done && !donePair && !Q.chaseL && !Q.chaseR) {
				bool sideL;
				if ((Q.szL < Q.szR || V1_DONE(Q.drR)) && !V1_DONE(Q.drL)) sideL = true;
				else if (!V1_DONE(Q.drR)) sideL = false;
				else { donePair = true; continue; }
				const uint32_t drMe = sideL ? Q.drL : Q.drR, drOther = sideL ? Q.drR : Q.drL;
				uint32_t& szMe = sideL ? Q.szL : Q.szR;
				uint32_t& szOther = sideL ? Q.szR : Q.szL;
				bool& delayedMe = sideL ? Q.delayedL : Q.delayedR;
				bool& delayedOther = sideL ? Q.delayedR : Q.delayedL;
				bool& chaseMe = sideL ? Q.chaseL : Q.chaseR;
				bool& chaseOther = sideL ? Q.chaseR : Q.chaseL;
				if (V1_DONE(drOther) && szOther == 0) { donePair = true; continue; }     /* no pair in this orientation */
				if (!dr_found(X, drMe)) BF_ADVANCE_TOP(X, drMe);
				if (dr_found(X, drMe)) {
					const uint32_t leaf = AW(drMe + CA_LAST);
					szMe += (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP));        /* (aligner.h:1216, 1398: a 32-bit tally in the 64-bit build too) */
					if (szOther == 0 && szMe > 3u) delayedMe = true;                     /* dontReconcile_: aligner.h:1233 */
					else {
						if (szMe > symCeil && szOther > symCeil) { donePair = true; continue; }
						if (delayedOther && szOther < szMe) {
							delayedOther = false; delayedMe = true; chaseOther = true;
							chase_range_of(drOther, sideL ? (doneFw ? qlen1 : qlen2) : (doneFw ? qlen2 : qlen1));
						} else {
							chaseMe = true;
							chase_range_of(drMe, sideL ? (doneFw ? qlen2 : qlen1) : (doneFw ? qlen1 : qlen2));
						}
					}
				}
			}
		}
#undef V1_DONE
	}
	bf_read_end(X, B, 2u);
}
#endif


/* ---- the wavefront automaton ----------------------------------------------------------------------------------------
 * bf_run_read / bf_run_pair above run one read from start to finish, and a wavefront that runs 64 of them side by side
 * spends its time where a few lanes are: in round 4's section profile (profiles/r4/fourth_call_best_sections.txt) a pass
 * over the streak loop served 6 of the 64 lanes on e_coli and 3 at hg19 scale, a pass over the SA walk 2 -- every loop's
 * trip count is some lane's, and a read's turns are the slowest read's.  Here the same pieces (la_*, adv_*,
 * ch_advance_piece, the runners' turns) are states of ONE loop per wavefront (bt_best_kernels.hip):
 *   * a HOT round takes every lane that is extending a branch or walking the suffix array one step on -- the front's
 *     record, one query position (its rank blocks fetched at one place in the code for all lanes), the end of a streak
 *     (curtail, split, queue), one LF step of a walk -- whichever leaf of whichever read the lane is on;
 *   * lanes that need anything else (a driver's second halves, the runner's turn -- reporting, the mate's window --, the
 *     next driver's first halves, a new read) wait until enough of them do, and a COLD sweep takes them through those
 *     states together; a lane takes its next read there, while the other lanes are wherever they are.
 * The statements a read goes through, and their order, are those of bf_run_read / bf_run_pair (which remain: the host
 * build's reference for this loop, tests/emu, and PairedBWAlignerV1's runner, which is not taken apart). */
enum { BA_TAKE = 0, BA_IDLE, BA_END, BA_LEAF_EXIT, BA_POST, BA_RUN, BA_PRE, BA_LEAF_ENTER,      /* cold */
       BA_FRONT, BA_STEP, BA_SEND, BA_CHASE };                                                 /* hot */
#define BA_IS_HOT(ph) ((ph) >= BA_FRONT)
#define BA_IS_PENDING(ph) ((ph) >= BA_LEAF_EXIT && (ph) <= BA_LEAF_ENTER)     /* cold, and not waiting for a read */

struct BfAuto {
	uint32_t phase;
	uint32_t kind;                /* 1 bf_run_read, 2 bf_run_pair */
	uint32_t live;                /* the loop is to be gone through (the read was long enough, the tree fitted) */
	uint32_t afterAdv;            /* the runner's turn goes on behind its BF_ADVANCE_TOP */
	bool done, chase;
	uint32_t drv, pairsFw, pairsRc, mmBuf, attempts;
	uint32_t c0[9];               /* the lane's op counters when the read began (a read that overflows is not tallied) */
	BfChase ch;
	BfAdvSt adv;
	/* The leaf being extended (la_enter .. la_exit: the queue's front and its 16-word branch record, the range, the seed's
	 * edits) is what the hot rounds read and write; the kernel gives every lane a place for it in LDS (bt_best_kernels.hip:
	 * lane-strided, an odd number of words apart, so that a wavefront's accesses to one member fall into different banks) --
	 * until round 6 it sat in scratch memory with the rest of the record.  (The host build points it at ordinary memory.) */
	BfLeafSt* leafp;
};
/* words from one lane's BfLeafSt to the next in LDS: the struct's size rounded up to an odd number of words */
#define BF_LEAF_STRIDE ((((uint32_t)sizeof(BfLeafSt) + 3u) / 4u) | 1u)

/* bf_run_read / bf_run_pair down to their loops */
BF_FNI void bf_auto_begin(BfLane& X, const BtBatchDev& B, uint32_t rd, BfAuto& S)
{
	BF_PT0(t_begin);
	S.live = 0; S.done = true; S.chase = false; S.attempts = 0; S.afterAdv = 0;
	S.c0[0] = X.c_lfex; S.c0[1] = X.c_lf2; S.c0[2] = X.c_lf1; S.c0[3] = X.c_chase; S.c0[4] = X.c_ftab;
	S.c0[5] = X.c_offs; S.c0[6] = X.c_rst; S.c0[7] = X.c_same; S.c0[8] = X.c_frames;
	bf_read_begin(X, B, rd);
	if (X.R[0].len < 4u || (S.kind != 1u && X.R[1].len < 4u)) { X.status |= BT_STF_SKIPPED; return; }
	S.live = 1;
	bf_chase_init(S.ch);
	S.drv = bf_build_tree(X);
	if (S.kind == 1u) {
		if (!X.ovf) { cost_set_query<0>(X, S.drv); S.done = dr_done(X, S.drv); }
	} else {
		const uint32_t maxLen = X.R[0].len > X.R[1].len ? X.R[0].len : X.R[1].len;
		S.pairsFw = bf_alloc(X, 3); S.pairsRc = bf_alloc(X, 3); S.mmBuf = bf_alloc(X, maxLen);
		if (!X.ovf) {
			AW(S.pairsFw) = AW(S.pairsFw + 1u) = AW(S.pairsFw + 2u) = 0; AW(S.pairsRc) = AW(S.pairsRc + 1u) = AW(S.pairsRc + 2u) = 0;
			cost_set_query<0>(X, S.drv);
			S.done = false;
		}
	}
	BF_PADD(BP_BEGIN, t_begin);
}
BF_FNI void bf_auto_end(BfLane& X, const BtBatchDev& B, BfAuto& S)
{
	bf_read_end(X, B, S.kind == 1u ? 1u : 2u);
	if (X.status & BT_STF_OVERFLOW) {
		X.c_lfex = S.c0[0]; X.c_lf2 = S.c0[1]; X.c_lf1 = S.c0[2]; X.c_chase = S.c0[3]; X.c_ftab = S.c0[4];
		X.c_offs = S.c0[5]; X.c_rst = S.c0[6]; X.c_same = S.c0[7]; X.c_frames = S.c0[8];
	}
}

/* the turns of bf_run_read's / bf_run_pair's loop, up to the one that needs the driver advanced (-> BA_PRE; the turn goes
 * on behind the advance: afterAdv), a piece of the SA walk (-> BA_CHASE) or nothing any more (-> BA_END) */
BF_FNI void bf_auto_run(BfLane& X, const BtBatchDev& B, BfAuto& S)
{
	BfChase& ch = S.ch;
	const uint32_t drv = S.drv;
	if (!S.live) { S.phase = BA_END; return; }
	if (S.kind == 1u) {
		if (S.afterAdv) { S.afterAdv = 0; if (dr_done(X, drv) && !dr_found(X, drv) && !S.chase) S.done = true; }
		for (;;) {
			if (S.done || X.ovf) { S.phase = BA_END; return; }
			if (S.chase) {
				if (ch.tidx == BF_NONE32 && !ch.done) { S.phase = BA_CHASE; return; }
				if (ch.tidx != BF_NONE32) {
					const uint32_t leaf = AW(drv + CA_LAST);
					S.done = bf_report_leaf(X, B, leaf, ch.tidx, ch.toff, 0, (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP) - 1u),
					                        !leaf_spec(X, leaf).mirror);
					ch.tidx = BF_NONE32;
				} else {
					S.chase = false;
					dr_set(X, drv, BF_F_FOUND, false);
					S.done = dr_done(X, drv);
				}
			}
			if (!S.done && !S.chase) {
				if (dr_found(X, drv)) {
					const uint32_t leaf = AW(drv + CA_LAST);
					const uint32_t cost = AW(leaf + LF_CURCOST) & 0xffffu;
					ch_set_top_bot(X, ch, AR(leaf + LF_CURTOP), AR(leaf + LF_CURBOT), leaf_spec(X, leaf).mirror, X.R[0].len);
					if (ch.tidx != BF_NONE32) {
						S.done = bf_report_leaf(X, B, leaf, ch.tidx, ch.toff, 0, (uint32_t)(AR(leaf + LF_CURBOT) - AR(leaf + LF_CURTOP) - 1u),
						                        !leaf_spec(X, leaf).mirror);
						ch.tidx = BF_NONE32;
					}
					if (!ch.done && !bf_irrelevant(X, cost)) S.chase = true;
					else dr_set(X, drv, BF_F_FOUND, false);
				} else {
					S.done = bf_irrelevant(X, dr_mincost(X, drv));
					if (!S.done) { S.afterAdv = 1; S.phase = BA_PRE; return; }
				}
				if (dr_done(X, drv) && !dr_found(X, drv) && !S.chase) S.done = true;
			}
		}
	}
	/* bf_run_pair */
	bool tail = S.afterAdv != 0;      /* behind the advance: the found check of the turn that asked for it */
	S.afterAdv = 0;
	for (;;) {
		if (!tail) {
			if (S.done || X.ovf) { S.phase = BA_END; return; }
			if (S.chase) {
				if (ch.tidx == BF_NONE32 && !ch.done) { S.phase = BA_CHASE; return; }
				if (ch.tidx != BF_NONE32) {
					/* resolveOutstanding (aligner.h:1849-1871) */
					const bool ret = bf_resolve_in_ref(X, B, AW(drv + CA_LAST), ch.tidx, ch.toff, S.pairsFw, S.pairsRc, S.mmBuf);
					if (++S.attempts > X.P->pairTries || ret) S.done = true;
					ch.tidx = BF_NONE32;
				} else {
					S.chase = false;
					S.done = dr_done(X, drv);
				}
			}
			if (S.done || S.chase) continue;
			if (dr_done(X, drv)) { S.done = true; continue; }
			S.done = bf_irrelevant(X, dr_mincost(X, drv));
			if (!S.done) { S.afterAdv = 1; S.phase = BA_PRE; return; }
		}
		tail = false;
		if (dr_found(X, drv)) {
			S.chase = true;
			dr_set(X, drv, BF_F_FOUND, false);
			const uint32_t leaf = AW(drv + CA_LAST);
			const BfSpec& sp = leaf_spec(X, leaf);
			ch_set_top_bot(X, ch, AR(leaf + LF_CURTOP), AR(leaf + LF_CURBOT), sp.mirror, X.R[sp.mate].len);
		}
	}
}

/* one hot round of one lane; sendOk: streaks that have ended are finished this round (the wavefront's gate) */
/* -> what the lane went through: 1 a step, 2 the end of a streak, 4 a piece of a walk (the host's wave model counts them) */
BF_FNI uint32_t bf_auto_hot(BfLane& X, BfAuto& S, bool sendOk)
{
	uint32_t did = 0;
	BF_PT0(t_hot);
	if (S.phase == BA_FRONT) { la_front(X, *S.leafp); S.phase = BA_STEP; }
	if (S.phase == BA_STEP) { BF_PT0(t_s); did |= 1u; if (!la_step(X, *S.leafp)) S.phase = BA_SEND; BF_PADD(BP_HSTEP, t_s); }
	if (S.phase == BA_SEND && sendOk) { BF_PT0(t_s); did |= 2u; S.phase = la_send(X, *S.leafp) ? BA_FRONT : BA_LEAF_EXIT; BF_PADD(BP_HSEND, t_s); }
	if (S.phase == BA_CHASE) { BF_PT0(t_s); did |= 4u; ch_advance_piece(X, S.ch); if (S.ch.tidx != BF_NONE32 || S.ch.done) S.phase = BA_RUN; BF_PADD(BP_HCHASE, t_s); }
	BF_PADD(BP_HOT, t_hot);
	return did;
}

/* one pass of the cold sweep for one lane; takeOk: lanes that wait for a read take one (take() -> its number, or
 * 0xffffffff when the batch has none left) */
template <class Take>
BF_FNI void bf_auto_cold(BfLane& X, const BtBatchDev& B, BfAuto& S, bool takeOk, Take take)
{
	BF_PT0(t_cold);
	if (S.phase == BA_END) { bf_auto_end(X, B, S); S.phase = BA_TAKE; }
	if (S.phase == BA_TAKE && takeOk) {
		BF_PT0(t_s);
		const uint32_t rd = take();
		if (rd == 0xffffffffu) S.phase = BA_IDLE;
		else { bf_auto_begin(X, B, rd, S); S.phase = BA_RUN; }
		BF_PADD(BP_CTAKE, t_s);
	}
	if (S.phase == BA_LEAF_EXIT) { BF_PT0(t_s); la_exit(X, *S.leafp); S.phase = BA_POST; BF_PADD(BP_CEXIT, t_s); }
	if (S.phase == BA_POST) { BF_PT0(t_s); adv_post(X, S.adv); S.phase = BA_RUN; BF_PADD(BP_CPOST, t_s); }
	if (S.phase == BA_RUN) { BF_PT0(t_s); bf_auto_run(X, B, S); BF_PADD(BP_CRUN, t_s); }
	if (S.phase == BA_PRE) {
		BF_PT0(t_s);
		if (!adv_pre(X, S.adv, S.drv)) S.phase = BA_RUN;             /* no second halves: the runner's turn goes on (next pass) */
		else S.phase = S.adv.leaf ? BA_LEAF_ENTER : BA_POST;
		BF_PADD(BP_CPRE, t_s);
	}
	if (S.phase == BA_LEAF_ENTER) S.phase = la_enter(X, *S.leafp, S.adv.leaf) ? BA_FRONT : BA_POST;
	BF_PADD(BP_COLD, t_cold);
}

#undef AW
#endif /* BT_BEST_H_ */
