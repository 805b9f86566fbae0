/*
 * bt_core.h -- the per-read search state machine (one GPU lane = one read).
 *
 * What it computes is what a reference worker thread computes for a read between GET_READ and
 * FINISH_READ (ebwt_search.cpp:923-961) in the default unpaired modes: the phase scripts
 * (search_exact.c, search_1mm_phase*.c, search_23mm_phase*.c, search_seeded_phase*.c), the
 * randomized greedy-DFS backtracker GreedyDFSRangeSource (ebwt_search_backtrack.h:237-1091,
 * 1118-1655), the SA walk + offset resolution (Ebwt::reportChaseOne / joinedToTextOff,
 * ebwt.h:2569-2755) and the per-read hit-sink policy (hit.h:969-985, 1201-1209).
 *
 * How it computes it is not the reference's way.  The reference recurses and touches memory
 * wherever the algorithm happens to need it.  Here every read is an explicit-stack automaton with
 * ONE rule: *a lane issues at most one memory request per round and never waits inside a round*.
 * Whenever the automaton needs data that is not in registers or LDS -- the next LF-mapping (rank
 * over one or two BWT rows), the next 16 bases of the read, the (top,bot) ranges of a backtrack
 * target, a popped frame record, a batch of (mask,quality) records to re-scan, an ftab entry, an
 * SA sample -- it describes the request (BtReq), returns, and is resumed in the next round with
 * the data (BtRes).  The caller -- the HIP kernel in bt_kernels.hip -- advances the 64 reads of a
 * wavefront in lock step, issues the requests of all lanes back to back and waits once; which
 * phase / frame / SA walk each individual read is in does not matter.  Stores are fire-and-forget.
 *
 *   lane state   : BtLane -- bit-packed, lives in VGPRs
 *   read window  : 16 bases + 16 qualities of the read around the current position (registers)
 *   frame stack  : one 48-byte record per backtrack level in HBM scratch, the most recently
 *                  pushed one also in LDS (a failed child pops it back without a fetch)
 *   range stack  : per visited query position the 4x(top,bot) ranges (32 B) and a 16-bit
 *                  (eliminated-chars mask | Phred<<8) record; compact -- a child frame starts
 *                  where its parent stopped
 *   seedlings    : packed partial alignments of the -n seed phases (ebwt_search_util.h:37-88)
 *
 * This header is plain C++ that compiles for gfx950 (hipcc) and for the host; the host build
 * exists only so the automaton can be unit-tested against the oracle without a GPU
 * (tests/emu).  The product library never runs it on the CPU.
 */
#ifndef BT_CORE_H_
#define BT_CORE_H_

#include "bt_rank.h"

/* ---- phase program: one BtStep per searcher invocation of a phase script ------------------ */
enum { BT_OC_ZERO = 0, BT_OC_PLEN = 1, BT_OC_S = 2, BT_OC_S3 = 3, BT_OC_S5 = 4 };
enum { BT_KIND_SEARCH = 0, BT_KIND_GEN = 1, BT_KIND_EXTEND = 2 };

struct BtStep {
	uint8_t mirror;          /* 0: index of the text, 1: mirror index                           */
	uint8_t readFw;          /* EbwtSearchParams::setFw                                         */
	uint8_t kind;            /* SEARCH | GEN (collect seedlings, setQlen(seed)) | EXTEND        */
	uint8_t reportExacts;
	uint8_t considerQuals;
	uint8_t halfAndHalf;
	uint8_t maq;
	uint8_t reportPartials;
	uint8_t oc[6];           /* setOffs(5depth,3depth,unrev,1rev,2rev,3rev) as BT_OC_* codes    */
	uint8_t pad[2];
	uint32_t qualThresh, maxBts;
};

#define BT_MAX_STEPS 12
struct BtProgram {
	int32_t  nsteps;
	int32_t  seeded;         /* -n mode: apply the phase-1 prologue filter                      */
	uint32_t seedLen;        /* -l (0xffffffff in -v modes: the "seed" is the whole read)       */
	uint32_t seedMms;
	uint32_t minLen;         /* -v: reads shorter than this are an error (BT_ST_TOOSHORT)       */
	uint32_t sinkN, sinkMax; /* NGoodHitSinkPerThread _n/_max (hit.h:937-985)                   */
	uint32_t sinkAll;
	BtStep   steps[BT_MAX_STEPS];
};

/* ---- per-lane scratch in HBM ------------------------------------------------------------- */
#define BT_FR_WORDS 12       /* 48-byte frame record = three 16-byte pieces                     */
enum {
	FR_W0 = 0,   /* depth | d<<11                                                                */
	FR_W1,       /* ham | lowAltQual<<16                                                         */
	FR_W2,       /* fu | f1<<11 | elcint<<22 | elignore<<24 | candValid<<25 | ccValid<<26 (LDS copy only) */
	FR_W3,       /* f2 | f3<<11 | cel1<<22 | cel2<<26                                            */
	FR_W4,       /* altNum | eligibleNum<<12 | low2<<24                                          */
	FR_W5,       /* cand | dcf<<11 | lmode<<22 | lt<<23 | lz<<25 | el<<26 (locus mode, below)        */
	FR_W6,       /* pi | pj<<11 | pel<<13                                                        */
	FR_EBASE,
	FR_ANCHOR,   /* locus mode: the frame's anchor (text offset of its row's suffix + depth)       */
	FR_MM,       /* mismatch chosen at this level: query offset | refc<<16                       */
	FR_L2        /* the second quality level: cand2 | num2<<11 | l2v<<23 (low2 rides in FR_W4, cel1 / cel2 in FR_W3) */
};
#ifndef BT_L2_TALLY
/* 1: a frame tallies a SECOND quality level while it steps forward (BtLane::low2) and its first re-scan costs nothing.  Off:
 * the tally needs a tenth word in the LDS top-of-stack copy, and with it the three-block build's 54 144 bytes of LDS no
 * longer fit three to a CU (LDS is handed out in 1 280-byte granules on gfx950: 3 x 43 > 128) -- round 6's third GPU call
 * measured 12.0 M reads/s for it against 15.7 M without, i.e. two blocks per CU; -3.3 % rounds do not pay for that */
#define BT_L2_TALLY 0
#endif
#if BT_L2_TALLY
#define BT_L2_RESET(L) do { (L).l2v = 0; (L).low2 = 0xff; (L).cel1 = 0; (L).cel2 = 0; } while (0)
#else
#define BT_L2_RESET(L) do { } while (0)
#endif
#define BT_TOS_WORDS (9 + BT_L2_TALLY)   /* FR_W0..FR_ANCHOR (and FR_L2, as word 9) travel to the LDS top-of-stack copy */
#define BT_CC_WORDS 9        /* LDS copy of the current backtrack candidate: tops[4], bots[4], record */
#define BT_LDS_WORDS (BT_CC_WORDS + BT_TOS_WORDS + BT_CC_WORDS)   /* per lane: candidate, top-of-stack, its candidate */
#define BT_LITE_LDS_WORDS (BT_CC_WORDS + BT_TOS_WORDS)            /* the 3-waves-per-SIMD build: candidate, top-of-stack */
#ifndef BT_LITE_CC
#define BT_LITE_CC 1         /* 0: round 5's 3-waves build, without the current frame's candidate cache (A/B) */
#endif

struct BtArena {
	/* arena bases and capacities (wave-uniform).  On the GPU this lives in LDS, one copy per
	 * workgroup: addresses are formed where they are used and nothing stays in registers. */
	uint32_t* frames;   /* [slot][frame][12]                                                     */
	uint32_t* pairs;    /* [slot][entry][8] rows (bt_row): tops ACGT, bots ACGT                  */
	uint16_t* meta;     /* [slot][entry] eliminated-chars mask | Phred<<8                        */
	uint64_t* pals;     /* [slot][palCap] seedlings                                              */
	uint32_t  frCap, entCap, palCap, pad;
};
struct BtScratch {
	const BtArena* a;
	uint32_t* tos;      uint32_t tosStride;  /* LDS, word w at tos[w*tosStride]: [0,9) the candidate's ranges +
	                                            record, [9,19) top-of-stack frame record, [19,28) that
	                                            frame's candidate */
	uint32_t  slot;
	uint32_t* tosRec;   /* the top-of-stack record region of `tos` (tos + 9*tosStride, or the whole of a
	                       smaller LDS allocation when the candidate caches are off) */
	uint32_t  noCC;     /* 0: both LDS candidate caches (the current frame's and the top-of-stack frame's); 1: none
	                       (choosing a target then always fetches its ranges); 2: the current frame's only -- the
	                       3-waves-per-SIMD build since round 6 (BT_LITE_CC): a frame popped back has to fetch its
	                       target's ranges again, a frame that fails where it stands does not */
	uint32_t  rlMax;    /* longest read this build's LDS copy holds (112, or 104 in the 3-waves-per-SIMD build) */
	uint32_t* rl;       /* LDS copy of the lane's whole read (reads of <= BT_RL_MAXLEN bases; RL builds of the
	                       automaton): word w at rl[w*tosStride]; [0,7) the bases, TWO bits each (16 to a word:
	                       what the text is packed as, so that a stretch of the read is compared with the text by
	                       XOR), [7,7+rlMax/4) the qualities, one byte each, bit 7 = "this base is an N" */
};
#define BT_RL_MAXLEN 112u
#define BT_RL_SEQ_WORDS 7u
#define BT_RL_WORDS 35u
#define BT_RL3_MAXLEN 104u       /* the 3-waves-per-SIMD build: 7 base words + 26 quality words */
#define BT_RL3_WORDS 33u

/* ---- batch-level arguments --------------------------------------------------------------- */
struct BtHitRec {            /* == bt_hit (include/bowtie_amd.h) */
	uint32_t tidx, toff, oms, mm_off;
	uint16_t cost, nmm;
	uint8_t  stratum, fw, pad[2];
};

struct BtBatchDev {
	const uint8_t*  seq;  const uint8_t* qual;  const uint16_t* len;  const uint32_t* seed;
	uint32_t n_reads, stride;               /* stride: multiple of 16 (rows are 16-byte aligned) */
	BtHitRec* hits; uint32_t hit_cap;
	uint32_t* n_hits; uint8_t* status;
	uint16_t* mm_pool; uint32_t mm_pool_cap; uint32_t* mm_pool_used;
	uint32_t* iters;                        /* optional [n_reads]: lock-step rounds the read took       */
	/* paired-end (bt_align_pairs): the second mates, same layout; NULL otherwise */
	const uint8_t*  seq2; const uint8_t* qual2; const uint16_t* len2; const uint32_t* seed2;
	uint32_t stride2, pad2;
};

/* Arguments split by temperature.  BtHot is passed by value (kernarg -> SGPRs) and holds only what
 * the per-position / per-SA-step code touches; BtCold lives in device memory and is read where
 * it is used (phase changes, reporting). */
struct BtHot {
	const uint8_t* blk[2];            /* the rank blocks (bt_rank.h) of the text index and of the mirror index */
	uint32_t zBlk[2], zPos[2];
	bt_row   fchr[2][5];
	const uint8_t* seq; const uint8_t* qual;
	uint32_t stride, n_reads;
#if BT_WIDE
	const uint64_t* segBase[2];       /* the wide build's rank blocks count from their segment's start (bt_rank.h) */
	uint32_t segShift, padW;
#endif
};
/* BtWarm: index geometry the automaton reads a few times per frame / SA walk.  On the GPU it sits
 * in LDS (one copy per workgroup) so that it occupies no scalar registers across the round loop. */
struct BtWarm {
	const bt_row* ftab[2];
	const bt_row* offs[2];
	bt_row   zOff[2], offMask[2];
	uint32_t offRate[2], ftabChars[2];
	bt_row   len[2];
	/* the locus image (bt_rank.h), all NULL / 0 when it was not built or is switched off */
	const BtU4*     loc[2];
	const uint32_t* rtxt[2];
	const uint16_t* walk[2];
	uint32_t locOn, pad;
	/* the jump table (bt_rank.h), NULL / 0 when the index has none */
	const uint32_t* jump[2];
	const uint16_t* jumpMeta[2];
	uint32_t jumpChars[2];
#if BT_WIDE
	bt_row   rowLim[2];               /* the last BWT row (BtIndexDev::rowLim) */
#endif
};
#if BT_WIDE
#define BT_WROWLIM() WSEL(rowLim)
#else
#define BT_WROWLIM() WSEL(len)
#endif
/* batches whose reads may be in flight at once (carry-over): a read rides along for at most BT_BATCH_RING - 2 launches.  16 until
 * round 6: at twelve launches of 12 M reads the hardest reads of a batch were not done, and every launch from the thirteenth on
 * ended with half a second of the machine waiting for them */
#define BT_BATCH_RING 64
struct BtCold {
	BtProgram  P;
	BtIndexDev ix[2];            /* [0] index of the text, [1] mirror index */
	BtBatchDev B;                /* the batch being searched (= ring[curBid])                             */
	uint32_t   curBid, pad;
	BtBatchDev ring[BT_BATCH_RING];   /* the batches reads still in flight belong to (carry-over): a read's results go
	                                     to ring[L.bid]                                                            */
};

#define BT_STF_SKIPPED   1u
#define BT_STF_HITCAP    2u
#define BT_STF_TOOSHORT  4u
#define BT_STF_OVERFLOW  8u     /* a per-read scratch capacity was exceeded; results invalid      */
#define BT_STF_MMPOOL    16u    /* mm_pool exhausted; hit stored without its mismatch list        */

/* ---- the one memory request of a round, and its answer -------------------------------------- */
enum { RQ_NONE = 0, RQ_RANK = 1, RQ_FETCH = 2 };
struct BtReq {
	uint32_t kind;          /* RQ_*                                                               */
	uint32_t n;             /* RANK: 1 or 2 rows; FETCH: 16-byte pieces at a (1..4)               */
	/* RANK : a = rowA, x = rowB; wchunk != 0xffff: also fetch 16-byte chunk `wchunk` of the lane's
	 *        read (bases -> res.q[3], qualities -> res.x): the next read window, one step ahead
	 * FETCH: a = address of n contiguous 16-byte pieces -> res.q[0..n)
	 *        x = address of an optional extra 16-byte piece -> res.x (0 = none) */
	uint64_t a, x;
	uint32_t wchunk;
	/* not a request: what this call's locus-mode steps decided by the text instead of going through -- mapLFEx steps in the
	 * high half, mapLF1 steps in the low half (each at most a read's length per call).  The caller adds them up (the kernel
	 * per lane in a register, flushed now and then: an LDS atomic per step costs more than the step) */
	uint32_t tally;
};
struct BtRes {
	BtU4 q[4];              /* RANK: q[0] = LF(rowA, ACGT), q[1] = LF(rowB, ACGT), q[2].x = BWT char at rowA
	                           (wide build: a quartet is four 64-bit rows = two pieces -- q[0..1] = LF(rowA, ACGT),
	                           q[2..3] = LF(rowB, ACGT), x.x = BWT char at rowA -- the layout of a fetched range-stack entry) */
	BtU4 x;
};
/* the two quartets of a rank answer / of a fetched range-stack entry, and single rows out of fetched pieces */
#if BT_WIDE
#define BT_RES_ROWL(res) ((res).x.x)
BT_HD bt_row bt_u4_row2(const BtU4& v, uint32_t k) { return (k & 1u) ? (((uint64_t)v.w << 32) | v.z) : (((uint64_t)v.y << 32) | v.x); }
BT_HD void bt_res_quartets(const BtRes& res, bt_row ta[4], bt_row tb[4])
{
	ta[0] = bt_u4_row2(res.q[0], 0); ta[1] = bt_u4_row2(res.q[0], 1); ta[2] = bt_u4_row2(res.q[1], 0); ta[3] = bt_u4_row2(res.q[1], 1);
	tb[0] = bt_u4_row2(res.q[2], 0); tb[1] = bt_u4_row2(res.q[2], 1); tb[2] = bt_u4_row2(res.q[3], 0); tb[3] = bt_u4_row2(res.q[3], 1);
}
/* row k of a table fetched as the 16-byte piece(s) that hold it */
BT_HD bt_row bt_piece_row(const BtU4& v, uint32_t k) { return bt_u4_row2(v, k); }
#else
#define BT_RES_ROWL(res) ((res).q[2].x)
BT_HD void bt_res_quartets(const BtRes& res, uint32_t ta[4], uint32_t tb[4])
{
	ta[0] = res.q[0].x; ta[1] = res.q[0].y; ta[2] = res.q[0].z; ta[3] = res.q[0].w;
	tb[0] = res.q[1].x; tb[1] = res.q[1].y; tb[2] = res.q[1].z; tb[3] = res.q[1].w;
}
#endif
#define BT_ENT_PIECES (2u * (uint32_t)sizeof(bt_row) / 4u)     /* 16-byte pieces of a range-stack entry: 2, wide 4 */
/* the locus-mode paths of the RL builds: not in the wide build (bt_rank.h, "the row type") */
#define BT_LOC(RL) ((RL) && !BT_WIDE)

enum {
	ST_IDLE = 0,
	/* fast states */
	ST_STEP_BEGIN, ST_STEP_LFDONE, ST_STEP_POST, ST_CHASE_CHECK, ST_CHASE_LFDONE, ST_WIN_DONE,
	ST_LOC_REC, ST_LOC_TXT, ST_STEP_LOC,      /* locus mode: a row's locus record / a text window arrived; a step decided by text */
	/* slow states */
	ST_PHASE_NEXT, ST_SEARCH_BEGIN, ST_FTABSEQ_DONE, ST_FTAB_DONE, ST_JUMP_DONE, ST_FRAME_ENTER, ST_BT_LOOP, ST_BT_PICK,
	ST_CANDSCAN, ST_CANDSCAN_DONE, ST_CHILD_RET, ST_RESCAN, ST_RESCAN_DONE, ST_FRAME_RETURN,
	ST_FRAME_FETCHED, ST_FELL_OFF, ST_RA_BEGIN, ST_ROW_BEGIN, ST_RESOLVE_DONE, ST_RA_END, ST_SEARCH_END,
	ST_ABORT
};
#define BT_IS_SLOW(st) ((st) >= ST_PHASE_NEXT)
enum { RC_STEP = 0, RC_CHILD, RC_FELL, RC_ENTRY };
enum { LFK_EX2 = 0, LFK_C2, LFK_LF1, LFK_CHASE };

/* op counters (bt_op_counts order) */
enum { CN_LFEX = 0, CN_LF2, CN_LF1, CN_CHASE, CN_FTAB, CN_OFFS, CN_RSTARTS, CN_FRAMES, CN_ITERS, CN_SAMEPAIR,
       CN_RESCAN, CN_CANDSCAN, CN_WROUNDS, CN_FETCH,
       /* locus mode: the reference's mapLFEx / mapLF1 / SA-walk steps that were decided by text comparison or by the dense
        * suffix array instead of being gone through (they are part of bt_op_counts' lfex / lf1 / chase all the same), the locus
        * records and the text windows fetched for it */
       CN_TLFEX, CN_TLF1, CN_TCHASE, CN_LOCREC, CN_TXTWIN,
       /* the jump table: the reference's mapLF (two rows) / mapLF1 steps behind a look-up, and how many of the former had both
        * rows in one side pair (part of bt_op_counts' lf2 / lf1 / same_pair all the same); the look-ups themselves */
       CN_JLF2, CN_JLF1, CN_JSAME, CN_JUMPS, CN_N };
#if defined(__HIP_DEVICE_COMPILE__)
/* one LDS atomic per wavefront: hipcc folds atomicAdd(p,1) of the active lanes into s_bcnt1 + one ds_add */
#define BT_COUNT(k) atomicAdd(&CNT[k], 1ull)
#define BT_COUNT_N(k, n) atomicAdd(&CNT[k], (unsigned long long)(n))
#define BT_COUNT_HOST(k)            /* tallied by the kernel with wave-uniform counters */
#else
#define BT_COUNT(k) (CNT[k]++)
#define BT_COUNT_N(k, n) (CNT[k] += (n))
#define BT_COUNT_HOST(k) (CNT[k]++)
#endif

/* Section timers for the profiling build (-DBT_PROFILE, scripts/prof_sections.py): wavefront
 * cycles (s_memtime) per section, accumulated in LDS.  No-ops in the product build. */
enum { PS_RESUME = 0, PS_SLOW, PS_WAIT, PS_RANK, PS_REFILL, PS_LOOP,
       PS_FELL_OFF, PS_RESOLVE_DONE, PS_RA_END, PS_FRAME_RETURN, PS_CHILD_RET, PS_RESCAN, PS_SEARCH_END, PS_PHASE_NEXT, PS_SEARCH_BEGIN, PS_FTABSEQ_DONE, PS_FTAB_DONE, PS_BT_LOOP, PS_CANDSCAN, PS_BT_PICK, PS_RA_BEGIN, PS_ROW_BEGIN, PS_FRAME_ENTER, PS_PASSES, PS_SINGLE_LFEX, PS_SINGLE_RUNS, PS_LOCUS, PS_LOC_MM, PS_LOC_TALLY, PS_LOCUS_PASSES, PS_RUN_ITERS, PS_N };
#if defined(BT_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define BT_PROF_T0(v) const unsigned long long v = __builtin_readcyclecounter()
#define BT_PROF_PASS() do { const unsigned long long ex_ = __ballot(1); if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex_)) atomicAdd(&CNT[CN_N + PS_PASSES], 1ull); } while (0)
#define BT_PROF_TICK(k) do { const unsigned long long ex_ = __ballot(1); if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex_)) atomicAdd(&CNT[CN_N + (k)], 1ull); } while (0)
#define BT_PROF_ADD(k, v) do { const unsigned long long ex_ = __ballot(1); \
	if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex_)) atomicAdd(&CNT[CN_N + (k)], __builtin_readcyclecounter() - (v)); } while (0)
#else
#define BT_PROF_T0(v)
#define BT_PROF_ADD(k, v)
#define BT_PROF_PASS()
#define BT_PROF_TICK(k)
#endif

struct BtLane {
	/* read */
	uint32_t rd, seed;
	uint64_t roff;                   /* rd * stride */
	uint32_t plen : 11, status : 8, step : 5, kind : 2, nmuts : 2, palIdxBefore : 1, hasN : 1;
	uint32_t nhits;
	uint32_t stored : 16, npals : 16;
	uint32_t palIdx : 16, iham : 8, mutnew0 : 2, mutnew1 : 2, mutnew2 : 2;
	uint32_t mutpos0 : 10, mutpos1 : 10, mutpos2 : 10;
	/* searcher (GreedyDFSRangeSource members) */
	uint32_t qlen : 11, mirror : 1, readFw : 1, rev : 1, reportExacts : 1, considerQuals : 1, halfAndHalf : 1,
	         maq : 1, reportPartials : 2, bailed : 1, nsFtab0 : 1;
	uint32_t d5 : 11, d3 : 11, bid : 6;          /* bid: the read's batch is C.ring[bid] (BT_BATCH_RING <= 64) */
	uint32_t unrev : 11, r1 : 11;
	uint32_t r2 : 11, r3 : 11;
	uint32_t qualThresh;
	uint32_t rnd, numBts;
	/* current frame (locals of backtrack(), ebwt_search_backtrack.h:363-455) */
	uint32_t sd : 7, depth : 11, d : 11;
	bt_row   top, bot;
	uint32_t ham : 16, lowAltQual : 8,
	         el : 6;                 /* locus mode: levels of the reference's recursion gone through without a frame of their own since the last real one (bt_loc_descend) */
	uint32_t fu : 11, f1 : 11, elcint : 2, elignore : 1, candValid : 1;
	uint32_t f2 : 11, f3 : 11,
	         cel1 : 4, cel2 : 4;     /* eliminated-sets of the deepest position of the eligible quality / of the second level (below) */
	uint32_t altNum : 12, eligibleNum : 12,
	         low2 : 8;               /* the SECOND-lowest quality among the frame's alternatives (round 6), with ... */
	uint32_t cand : 11, scanCb : 16;         /* scanCb: next chunk (8 records) of a running frame scan */
	/* ... its untried substitutions, its deepest position, and whether the three are known.  While a frame steps forward it
	 * tallies TWO quality levels instead of one: when the lowest is used up, the re-scan of the frame's records
	 * (ebwt_search_backtrack.h:1004-1058: next-lowest quality, its count, its deepest position) is then already known --
	 * nothing touched those records since -- and costs neither its fetch rounds nor its walk (43 re-scans per read at hg19
	 * scale, 9 % of the wavefronts' time in round 5's section profile).  The level after that is found by the scan as before. */
#if BT_L2_TALLY
	uint32_t cand2 : 11, num2 : 12, l2v : 1;
#endif
	uint32_t ebase;
	/* per-position temporaries that live across the wait + control */
	uint32_t c : 3, q : 8, lfk : 2, fl_alt : 1, fl_elig : 1, fl_over : 1, ret : 1, ra_cont : 2,
	         /* locus mode (RL builds, when the index has its locus image): the frame's range is one BWT row whose place in the
	          * text is known -- `top` holds the ANCHOR (text offset of the row's suffix + depth: the base compared at depth
	          * d is T[anchor - d - 1]) and bot = top + 1 (top: the range became empty) -- from depth `dcf` on.  The frame's
	          * range-stack entries are its positions below dcf plus ONE for its last position (bt_ent) */
	         lmode : 1, dcf : 11;
	/* pending backtrack target */
	uint32_t pi : 11, pj : 2, btham : 16;
	uint32_t pel : 4, tosFrame : 7, tosValid : 1, ccValid : 1, wpf : 1, cchunk : 8,   /* cchunk: the cached 16-byte chunk of the read (register-window build), 0xff = none */
	         lt : 2, lz : 1,      /* locus mode: the text's base at the frame's last position (its one alternative there); lz: there is none (text start) */
	         ra_l : 1;            /* the alignment being reported comes from locus mode: ra_top is an anchor, not a row */
	/* report */
	uint32_t ra_sd : 7, ra_stratum : 7, ra_cost : 16;
	bt_row   ra_top, ra_bot, ra_r, ra_i;
	bt_row   crow;
	uint32_t cjumps;
	uint32_t iters;
	/* register window over the read: 16 bases + 16 quals around the current position */
	uint32_t state;                  /* ST_*: in a word of its own -- every guard of the sweep tests it */
	uint32_t cs0, cs1, cs2, cs3, cq0, cq1, cq2, cq3;
};

/* ---- small helpers ----------------------------------------------------------------------- */
BT_HD uint32_t bt_rnd_u32(BtLane& L)                       /* RandomSource::nextU32, random_source.h:45-54 */
{
	uint32_t ret;
	L.rnd = 1664525u * L.rnd + 1013904223u;
	ret = L.rnd >> 16;
	L.rnd = 1664525u * L.rnd + 1013904223u;
	ret ^= L.rnd;
	return ret;
}
BT_HD uint32_t bt_mm_penalty(uint32_t maq, uint32_t q)     /* qual.h:61-67, qual.cpp:4-32 */
{
	if (!maq) return q;
	if (q < 5) return 0;
	if (q < 15) return 10;
	if (q < 25) return 20;
	return 30;
}
BT_HD uint32_t bt_apply_muts(const BtLane& L, uint32_t i, uint32_t c)
{
	if (L.nmuts > 0) {
		if (i == L.mutpos0) c = L.mutnew0;
		if (L.nmuts > 1 && i == L.mutpos1) c = L.mutnew1;
		if (L.nmuts > 2 && i == L.mutpos2) c = L.mutnew2;
	}
	return c;
}
/* query char / quality at index i of the string setQuery selected (ebwt_search_backtrack.h:90-140),
 * with the seedling mutations applied (:1368-1382).  RL: from the lane's LDS copy of the read;
 * otherwise a direct (synchronous) global load, used on rare paths only. */
/* the LDS copy: base j as a 2-bit code (an N reads as 0 here), its quality byte with the N flag in bit 7 */
BT_HD uint32_t bt_rl_base2(const BtScratch& S, uint32_t j)
{
	return (S.rl[(j >> 4) * S.tosStride] >> ((j & 15u) * 2u)) & 3u;
}
BT_HD uint32_t bt_rl_qbyte(const BtScratch& S, uint32_t j)
{
	return (S.rl[(BT_RL_SEQ_WORDS + (j >> 2)) * S.tosStride] >> ((j & 3u) * 8u)) & 0xffu;
}
template <bool RL>
BT_HD uint32_t bt_qry(const BtLane& L, const BtHot& H, const BtScratch& S, uint32_t i)
{
	uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	uint32_t c;
	if (RL) { c = bt_rl_base2(S, j); if (L.hasN && (bt_rl_qbyte(S, j) & 0x80u)) c = 4u; }
	else c = (uint32_t)BT_GP(const uint8_t, H.seq)[L.roff + j];
	if (!L.readFw && c < 4u) c ^= 3u;
	return bt_apply_muts(L, i, c);
}
template <bool RL>
BT_HD uint32_t bt_qual(const BtLane& L, const BtHot& H, const BtScratch& S, uint32_t i)
{
	uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	uint32_t v = RL ? (bt_rl_qbyte(S, j) & 0x7fu) : (uint32_t)BT_GP(const uint8_t, H.qual)[L.roff + j];
	return v >= 33u ? v - 33u : 0u;
}
/* copy the lane's read into its LDS slot (RL): 16 bases + 16 qualities per step */
BT_HD void bt_rl_store_chunk(const BtScratch& S, uint32_t base, const BtU4& sv, const BtU4& qv)
{
	const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
	uint32_t p = 0;
	BT_UNROLL
	for (int k = 0; k < 4; k++)
		p |= ((w[k] & 3u) | ((w[k] >> 6) & 0xcu) | ((w[k] >> 12) & 0x30u) | ((w[k] >> 18) & 0xc0u)) << (8 * k);
	const uint32_t ts = S.tosStride;
	/* the last chunk of a 104-base layout (26 quality words) is half a chunk: its upper half is padding */
	const bool full = base + 16u <= S.rlMax;
	S.rl[(base >> 4) * ts] = p;
	const uint32_t qb = BT_RL_SEQ_WORDS + (base >> 2);
	/* an N (code 4) has bit 2 set: it goes to bit 7 of the base's quality byte (qualities are ASCII, below 128) */
	S.rl[(qb + 0u) * ts] = (qv.x & 0x7f7f7f7fu) | ((w[0] & 0x04040404u) << 5); S.rl[(qb + 1u) * ts] = (qv.y & 0x7f7f7f7fu) | ((w[1] & 0x04040404u) << 5);
	if (full) { S.rl[(qb + 2u) * ts] = (qv.z & 0x7f7f7f7fu) | ((w[2] & 0x04040404u) << 5); S.rl[(qb + 3u) * ts] = (qv.w & 0x7f7f7f7fu) | ((w[3] & 0x04040404u) << 5); }
}
BT_HD void bt_rl_load(const BtLane& L, const BtHot& H, const BtScratch& S)
{
	BT_NOUNROLL
	for (uint32_t base = 0; base < L.plen; base += 16u)
		bt_rl_store_chunk(S, base, bt_ld4(H.seq + L.roff + base), bt_ld4(H.qual + L.roff + base));
}
BT_HD uint32_t bt_sel4(uint32_t k, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
	uint32_t lo = (k & 1u) ? w1 : w0, hi = (k & 1u) ? w3 : w2;
	return (k & 2u) ? hi : lo;
}
BT_HD uint32_t bt_u4_word(const BtU4& v, uint32_t k) { return bt_sel4(k, v.x, v.y, v.z, v.w); }
#if BT_WIDE
BT_HD bt_row bt_selr4(uint32_t k, bt_row w0, bt_row w1, bt_row w2, bt_row w3)
{
	bt_row lo = (k & 1u) ? w1 : w0, hi = (k & 1u) ? w3 : w2;
	return (k & 2u) ? hi : lo;
}
/* LF(rowA, character k) of a rank answer */
BT_HD bt_row bt_res_lfa(const BtRes& res, uint32_t k) { return (k & 2u) ? bt_u4_row2(res.q[1], k) : bt_u4_row2(res.q[0], k); }
#else
#define bt_selr4 bt_sel4
BT_HD uint32_t bt_piece_row(const BtU4& v, uint32_t k) { return bt_u4_word(v, k & 3u); }
BT_HD uint32_t bt_res_lfa(const BtRes& res, uint32_t k) { return bt_u4_word(res.q[0], k); }
#endif
BT_HD uint32_t bt_u4_byte(const BtU4& v, uint32_t b) { return (bt_u4_word(v, (b >> 2) & 3u) >> ((b & 3u) * 8u)) & 0xffu; }
/* the 16-bit record k (0..7) of a fetched chunk of the (mask,quality) array */
BT_HD uint32_t bt_u4_meta(const BtU4& v, uint32_t k)
{
	const uint32_t w = bt_u4_word(v, (k >> 1) & 3u);
	return (k & 1u) ? (w >> 16) : (w & 0xffffu);
}

#define FRW(f, w) BT_GP(uint32_t, S.a->frames)[((uint64_t)S.slot * S.a->frCap + (f)) * BT_FR_WORDS + (w)]
#define BT_ENT_WORDS (8u * (uint32_t)sizeof(bt_row) / 4u)        /* 32-bit words of a range-stack entry: 8 rows */
#define PT(e, c) BT_GP(bt_row, S.a->pairs)[((uint64_t)S.slot * S.a->entCap + (e)) * 8u + (c)]
#define PT4(e) (S.a->pairs + ((uint64_t)S.slot * S.a->entCap + (e)) * BT_ENT_WORDS)                       /* address of tops[4] */
#define PB4(e) (S.a->pairs + ((uint64_t)S.slot * S.a->entCap + (e)) * BT_ENT_WORDS + BT_ENT_WORDS / 2u)   /* address of bots[4] */
#define PB(e, c) BT_GP(bt_row, S.a->pairs)[((uint64_t)S.slot * S.a->entCap + (e)) * 8u + 4u + (c)]
/* a quartet of rows to an entry's tops / bots; a rank answer's two quartets to an entry */
#if BT_WIDE
BT_HD void bt_store_quartet(uint32_t* dst, bt_row a, bt_row b, bt_row c, bt_row d)
{
	BtU4 v; v.x = (uint32_t)a; v.y = (uint32_t)(a >> 32); v.z = (uint32_t)b; v.w = (uint32_t)(b >> 32); bt_st4(dst, v);
	v.x = (uint32_t)c; v.y = (uint32_t)(c >> 32); v.z = (uint32_t)d; v.w = (uint32_t)(d >> 32); bt_st4(dst + 4, v);
}
#define BT_STORE_ENTRY(e, res) do { uint32_t* pe_ = PT4(e); bt_st4(pe_, (res).q[0]); bt_st4(pe_ + 4, (res).q[1]); bt_st4(pe_ + 8, (res).q[2]); bt_st4(pe_ + 12, (res).q[3]); } while (0)
#else
BT_HD void bt_store_quartet(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
	BtU4 v; v.x = a; v.y = b; v.z = c; v.w = d; bt_st4(dst, v);
}
#define BT_STORE_ENTRY(e, res) do { bt_st4(PT4(e), (res).q[0]); bt_st4(PB4(e), (res).q[1]); } while (0)
#endif
#define META(e) BT_GP(uint16_t, S.a->meta)[(uint64_t)S.slot * S.a->entCap + (e)]
#define META_MASK(e) BT_GP(uint8_t, S.a->meta + (uint64_t)S.slot * S.a->entCap + (e))[0]   /* low byte: the eliminated-set */
#define PALS(k) BT_GP(uint64_t, S.a->pals)[(uint64_t)S.slot * S.a->palCap + (k)]
#define IXSEL(f) (L.mirror ? IX[1].f : IX[0].f)      /* cold: device memory */
#define HSEL(f) (L.mirror ? H.f[1] : H.f[0])          /* hot: scalar registers */
#define WSEL(f) (L.mirror ? W.f[1] : W.f[0])          /* warm: LDS */
#define HFCHR(k) (L.mirror ? H.fchr[1][k] : H.fchr[0][k])
/* the mismatch chosen at backtrack level i: query offset | refc<<16 (the FR_MM word of the frame record) */
#define BT_FR_MM(i) FRW((i), FR_MM)
/* a block that emits its request leaves the lane in a state no LATER block of the sweep takes (its own *_DONE /
 * *_FETCHED state) -- except the three that are entered straight from the block before them with the request pending */
#define ST_IS(x) (L.state == (x))
#define ST_IS_NOREQ(x) (L.state == (x) && req.kind == RQ_NONE)

#define BT_REQ_RANK1(R) do { req.kind = RQ_RANK; req.n = 1; req.a = (R); } while (0)
#define BT_REQ_RANK2(RA, RB) do { req.kind = RQ_RANK; req.n = 2; req.a = (RA); req.x = (RB); } while (0)
#define BT_REQ_FETCH(PA, N, PX) do { req.kind = RQ_FETCH; req.n = (N); req.a = (uint64_t)(uintptr_t)(const void*)(PA); req.x = (uint64_t)(uintptr_t)(const void*)(PX); } while (0)

BT_HD uint32_t bt_off_code(uint32_t plen, uint32_t qs, uint32_t code)
{
	return code == BT_OC_ZERO ? 0u : code == BT_OC_PLEN ? plen : code == BT_OC_S ? qs :
	       code == BT_OC_S3 ? (qs >> 1) : ((qs >> 1) + (qs & 1u));
}

/* Range-stack entry of position i of the current frame.  A frame in locus mode has entries for its positions below dcf
 * (they were gone through in row space) and ONE more, for the position it stops at: the positions between are stretches
 * that matched the text, which no scan and no backtrack ever looks at (a matching position has nothing to substitute). */
BT_HD uint32_t bt_ent(const BtLane& L, uint32_t i)
{
	return L.ebase + ((L.lmode && i >= L.dcf ? (uint32_t)L.dcf : i) - L.depth);
}
/* ... and back: the position of the entry `t` places above the entry of position `from` */
BT_HD uint32_t bt_ent_pos(const BtLane& L, uint32_t from, uint32_t t)
{
	return (L.lmode && from + t >= L.dcf) ? (uint32_t)L.d : from + t;
}

/* current position through the register window; false if the window has to be fetched first */
BT_HD bool bt_window_get(const BtLane& L, uint32_t i, uint32_t* c_out, uint32_t* q_out)
{
	const uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	if ((j >> 4) != L.cchunk) return false;
	const uint32_t k = (j >> 2) & 3u, sh = (j & 3u) * 8u;
	uint32_t c = (bt_sel4(k, L.cs0, L.cs1, L.cs2, L.cs3) >> sh) & 0xffu;
	uint32_t v = (bt_sel4(k, L.cq0, L.cq1, L.cq2, L.cq3) >> sh) & 0xffu;
	if (!L.readFw && c < 4u) c ^= 3u;
	*c_out = bt_apply_muts(L, i, c);
	*q_out = v >= 33u ? v - 33u : 0u;
	return true;
}

/* the same from the LDS copy of the read (RL): always there */
BT_HD void bt_read_get(const BtLane& L, const BtScratch& S, uint32_t i, uint32_t* c_out, uint32_t* q_out)
{
	const uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	uint32_t c = bt_rl_base2(S, j);
	uint32_t v = bt_rl_qbyte(S, j);
	if (v & 0x80u) { c = 4u; v &= 0x7fu; }
	if (!L.readFw && c < 4u) c ^= 3u;
	*c_out = bt_apply_muts(L, i, c);
	*q_out = v >= 33u ? v - 33u : 0u;
}


/* hhCheckTop (ebwt_search_backtrack.h:1200-1275) */
BT_HD bool bt_hh_check_top(const BtLane& L, const BtScratch& S, uint32_t d)
{
	if (d == L.d5) {
		if (L.sd == 0) return false;
	} else if (d == L.d3) {
		if (L.r3 == L.r2) {
			if (L.sd < 2) return false;
		} else {
			uint32_t lo = 0;
			BT_NOUNROLL
			for (uint32_t i = 0; i < L.sd; i++) {
				uint32_t dd = L.qlen - (BT_FR_MM(i) & 0xffffu) - 1u;
				if (dd >= L.d5 && dd < L.d3) lo++;
			}
			if (lo == 0) return false;
		}
	}
	return true;
}

/* reportPartial (ebwt_search_backtrack.h:1571-1655) */
BT_HD void bt_report_partial(BtLane& L, const BtScratch& S, uint32_t sd)
{
	uint64_t p0 = 0xffff, p1 = 0xffff, p2 = 0xffff, c0 = 3, c1 = 3, c2 = 3;
	if (sd > 0) { uint32_t mm = BT_FR_MM(0); p0 = mm & 0xffffu; c0 = (mm >> 16) & 3u; }
	if (sd > 1) { uint32_t mm = BT_FR_MM(1); p1 = mm & 0xffffu; c1 = (mm >> 16) & 3u; }
	if (sd > 2) { uint32_t mm = BT_FR_MM(2); p2 = mm & 0xffffu; c2 = (mm >> 16) & 3u; }
	uint64_t al = p0 | (p1 << 16) | (p2 << 32) | (c0 << 48) | (c1 << 50) | (c2 << 52) | (0xffull << 54) | (3ull << 62);
	if (L.npals < S.a->palCap) { PALS(L.npals) = al; L.npals = L.npals + 1u; }
	else L.status = L.status | BT_STF_OVERFLOW;
}

/* Start read `rd`: the worker-loop prologue (ebwt_search.cpp:1675-1683, 2167-2175, 2572-2584;
 * search_seeded_phase1.c:17-44). */
template <bool RL>
BT_HD void bt_lane_start(BtLane& L, const BtProgram& P, const BtHot& H, const BtCold& C, const BtScratch& S, uint32_t rd)
{
	L.rd = rd;
	L.roff = (uint64_t)rd * H.stride;
	L.plen = BT_GP(const uint16_t, C.B.len)[rd];
	L.seed = BT_GP(const uint32_t, C.B.seed)[rd];
	L.nhits = 0; L.stored = 0; L.status = 0;
	L.step = 31; L.npals = 0; L.palIdx = 0; L.nmuts = 0; L.palIdxBefore = 0;
	L.mirror = 0; L.readFw = 1; L.rev = 0;
	L.cchunk = 0xffu;
	L.iters = 0; L.tosValid = 0; L.ccValid = 0; L.bid = C.curBid;
	L.state = ST_PHASE_NEXT;
	if (RL && L.plen > S.rlMax) {
		/* longer than this build keeps in LDS (the caller's bt_ctx_set_max_read_len promise did not hold): not
		 * searched, flagged */
		L.plen = 0; L.status = BT_STF_OVERFLOW;
	}
	const uint32_t plen = L.plen;
	const uint32_t qs = plen < P.seedLen ? plen : P.seedLen;
	/* one pass of 16-byte loads over the read: does it contain an N at all (almost never), and
	 * how many in the seed (search_seeded_phase1.c:24-44) */
	uint32_t nsAll = 0, nsSeed = 0;
	BT_NOUNROLL
	for (uint32_t base = 0; base < plen; base += 16u) {
		const BtU4 v = bt_ld4(H.seq + L.roff + base);
		if (RL) bt_rl_store_chunk(S, base, v, bt_ld4(H.qual + L.roff + base));
		/* bit i of nb = byte i of the chunk is an N (code 4): exact zero-byte test on v ^ 0x04.., then
		 * the four flag bits of each word are gathered with one multiply */
		uint32_t nb = 0;
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		BT_UNROLL
		for (int k = 0; k < 4; k++) {
			const uint32_t t = w[k] ^ 0x04040404u;
			const uint32_t z = (~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u) >> 7;   /* 0x01 per N byte */
			nb |= ((z * 0x00204081u) >> 21 & 0xfu) << (4 * k);
		}
		const uint32_t inRead = plen - base >= 16u ? 0xffffu : ((1u << (plen - base)) - 1u);
		const uint32_t inSeed = qs <= base ? 0u : (qs - base >= 16u ? 0xffffu : ((1u << (qs - base)) - 1u));
		nsAll += (uint32_t)__builtin_popcount(nb & inRead);
		nsSeed += (uint32_t)__builtin_popcount(nb & inSeed);
	}
	L.hasN = nsAll > 0 ? 1u : 0u;
	if (P.seeded) {
		if (plen < 4u || nsSeed > P.seedMms) { L.status = L.status | BT_STF_SKIPPED; L.step = (uint32_t)P.nsteps - 1u; }
	} else if (plen < P.minLen) {
		L.status = L.status | BT_STF_TOOSHORT; L.step = (uint32_t)P.nsteps - 1u;
	} else if (plen == 0) {
		L.step = (uint32_t)P.nsteps - 1u;          /* nothing to search: unaligned */
	}
}

/* FINISH_READ: publish the sink counters (hit.h:741-786); the hit slots were written as found. */
BT_HD void bt_lane_finish(BtLane& L, const BtBatchDev& B)
{
	BT_GP(uint32_t, B.n_hits)[L.rd] = L.nhits;
	BT_GP(uint8_t, B.status)[L.rd] = (uint8_t)L.status;
	if (B.iters) B.iters[L.rd] = L.iters;
	L.state = ST_IDLE;
}

/* Ebwt::report + EbwtSearchParams::reportHit + sink (ebwt.h:2635-2682, 1288-1405; hit.h:969-985).
 * Returns true iff the sink says stop. */
BT_HD bool bt_report_hit(BtLane& L, const BtProgram& P, uint32_t ixfw, const BtScratch& S,
                         const BtBatchDev& B, uint32_t tidx, uint32_t toff)
{
	L.nhits++;
	if (L.nhits > P.sinkMax) return true;
	if (L.stored < B.hit_cap) {
		BtHitRec h;
		h.tidx = tidx; h.toff = toff; h.oms = (uint32_t)(L.ra_bot - L.ra_top - 1u);
		h.cost = (uint16_t)L.ra_cost; h.stratum = (uint8_t)L.ra_stratum; h.fw = (uint8_t)L.readFw;
		h.pad[0] = h.pad[1] = 0;
		const uint32_t nmm = L.ra_sd + L.nmuts;
		h.nmm = (uint16_t)nmm; h.mm_off = 0;
		if (nmm > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
			uint32_t off = atomicAdd(B.mm_pool_used, nmm);
#else
			uint32_t off = *B.mm_pool_used; *B.mm_pool_used += nmm;
#endif
			if (off + nmm <= B.mm_pool_cap) {
				h.mm_off = off;
				const bool flip = (ixfw != 0) != (L.readFw != 0);
				auto mm = BT_GP(uint16_t, B.mm_pool + off);
				BT_NOUNROLL
				for (uint32_t i = 0; i < nmm; i++) {
					uint32_t pos, refc;
					if (i < L.ra_sd) { uint32_t v = BT_FR_MM(i); pos = v & 0xffffu; refc = (v >> 16) & 3u; }
					else {
						const uint32_t k = i - L.ra_sd;
						pos = k == 0 ? L.mutpos0 : k == 1 ? L.mutpos1 : L.mutpos2;
						refc = k == 0 ? L.mutnew0 : k == 1 ? L.mutnew1 : L.mutnew2;
					}
					if (flip) pos = L.qlen - pos - 1u;
					uint16_t e = (uint16_t)(pos | (refc << 12));
					/* Hit::mms is a bitset: keep the list ordered by position */
					int j = (int)i - 1;
					while (j >= 0 && (mm[j] & 0x3ffu) > (e & 0x3ffu)) { mm[j + 1] = mm[j]; j--; }
					mm[j + 1] = e;
				}
			} else {
				h.nmm = 0; L.status = L.status | BT_STF_MMPOOL;
			}
		}
		{
			/* 24-byte record: one 16-byte and two 4-byte global stores */
			static_assert(sizeof(BtHitRec) == 24, "bt_hit layout");
			uint32_t hw[6];
			__builtin_memcpy(hw, &h, 24);
			uint32_t* dst = (uint32_t*)(B.hits + ((uint64_t)L.rd * B.hit_cap + L.stored));
			BtU4 q; q.x = hw[0]; q.y = hw[1]; q.z = hw[2]; q.w = hw[3];
			bt_st4(dst, q);
			BT_GP(uint32_t, dst)[4] = hw[4]; BT_GP(uint32_t, dst)[5] = hw[5];
		}
		L.stored = L.stored + 1u;
	} else if (L.stored < P.sinkN) {
		L.status = L.status | BT_STF_HITCAP;
	}
	if (P.sinkAll) return false;
	if (L.nhits == P.sinkN && (P.sinkMax == 0xffffffffu || P.sinkMax < P.sinkN)) return true;
	return false;
}

/* Begin reportAlignment (ebwt_search_backtrack.h:1455-1513) for `sd` mismatches on [top,bot). */
#define BT_GOTO_RA(SD, TOP, BOT, COST, CONT, LOC) \
	do { L.ra_sd = (SD); L.ra_top = (TOP); L.ra_bot = (BOT); L.ra_cost = (COST); L.ra_cont = (CONT); L.ra_l = (LOC) ? 1u : 0u; \
	     L.state = ST_RA_BEGIN; } while (0)

/* A frame scan (re-scan for the next eligible quality, or search for the deepest remaining target)
 * walks the frame's (mask,quality) records from the deepest position down, 32 records = four
 * 16-byte chunks = one fetch per round.  Request the next batch: chunks lo..scanCb. */
BT_HD void bt_scan_request(const BtLane& L, const BtScratch& S, uint32_t c_lo, BtReq& req)
{
	const uint32_t hi = L.scanCb, lo = hi >= c_lo + 3u ? hi - 3u : c_lo;
	BT_REQ_FETCH(&META((uint64_t)lo * 8u), hi - lo + 1u, nullptr);
}

/* One fetched piece (eight 16-bit records, the deepest last) of a frame scan.  Re-scan (:1004-1058): the reference's
 * walk from the deepest record down, restarting its tallies whenever a strictly lower quality shows up -- low = the
 * lowest affordable quality so far, num = untried substitutions at that quality, cnd / cel = the deepest record of that
 * quality (relative index t = e - e_lo) and its eliminated-set.  t0 = first record of the piece - e_lo; records with
 * t outside [0, span] are not part of the scan. */
struct BtScanAcc { uint32_t low, num, cnd, cel; };
BT_HD void bt_rescan_piece(const BtU4& q, uint32_t t0, uint32_t span, uint32_t qlim, BtScanAcc& a)
{
	const uint32_t w[4] = {q.x, q.y, q.z, q.w};
	BT_UNROLL
	for (int k = 7; k >= 0; k--) {
		const uint32_t el = (w[k >> 1] >> ((k & 1) * 16)) & 15u, kq = (w[k >> 1] >> ((k & 1) * 16 + 8)) & 0xffu;
		const uint32_t t = t0 + (uint32_t)k;
		const uint32_t key = (t <= span && el != 15u && kq <= qlim) ? kq : 0xffffffffu;
		const bool lt = key < a.low;
		a.low = lt ? key : a.low; a.num = lt ? 0u : a.num; a.cnd = lt ? t : a.cnd; a.cel = lt ? el : a.cel;
		a.num += (key == a.low) ? 4u - (uint32_t)__builtin_popcount(el) : 0u;
	}
}
/* Target scan (:767-812): the deepest record that still has an untried substitution of the eligible quality */
BT_HD void bt_candscan_piece(const BtU4& q, uint32_t t0, uint32_t span, uint32_t lowq, bool anyq, uint32_t& cnd)
{
	const uint32_t w[4] = {q.x, q.y, q.z, q.w};
	BT_UNROLL
	for (int k = 0; k < 8; k++) {                 /* shallow to deep: the last hit stays */
		const uint32_t el = (w[k >> 1] >> ((k & 1) * 16)) & 15u, kq = (w[k >> 1] >> ((k & 1) * 16 + 8)) & 0xffu;
		const uint32_t t = t0 + (uint32_t)k;
		cnd = (t <= span && el != 15u && (kq == lowq || anyq)) ? t : cnd;
	}
}


/* ---- locus mode: the pieces ---------------------------------------------------------------------------------------------
 * (RL builds.)  Reference behaviour replaced: the mapLF1 / one-row mapLFEx steps of GreedyDFSRangeSource::backtrack
 * (ebwt_search_backtrack.h:544-566 with ebwt.h:2334-2380, 2494-2512) from the step at which the range is one row [top, top+1).
 * That row's suffix starts at SA[top] in the joined text; extending it by a base c succeeds iff the text's base to the left is
 * c, and the new row's suffix starts one to the left.  So with anchor = SA[top] + d the base met at depth d' >= d is
 * T[anchor - d' - 1] (none if anchor == d': the '$' row, where mapLF1 gives OFF_MASK and mapLFEx four empty ranges), the
 * mapLFEx quartet there is "the text's base has the one row, the other three are empty", and everything the reference does
 * between two depths at which read and text differ -- elims that allow nothing, no alternatives, no eligibility changes --
 * leaves no trace.  The automaton therefore jumps from event to event: the next depth at which read and text differ, the
 * end of the read, the half-and-half boundaries. */
BT_HD uint32_t bt_rev2(uint32_t x)                          /* the sixteen 2-bit groups of x in reverse order */
{
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t r = __builtin_bitreverse32(x);
#else
	uint32_t r = x;
	r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1); r = ((r >> 2) & 0x33333333u) | ((r & 0x33333333u) << 2);
	r = ((r >> 4) & 0x0f0f0f0fu) | ((r & 0x0f0f0f0fu) << 4); r = ((r >> 8) & 0x00ff00ffu) | ((r & 0x00ff00ffu) << 8);
	r = (r >> 16) | (r << 16);
#endif
	return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
/* the query's characters at depths dd .. dd+15, two bits each, depth dd lowest (positions beyond the query hold garbage) */
BT_HD uint32_t bt_loc_qword(const BtLane& L, const BtScratch& S, uint32_t dd)
{
	/* depth dd is query index qlen-1-dd, which is read position j = rev ? plen-qlen+dd : qlen-1-dd: ascending with the
	 * depth in the one case, descending in the other (then the stretch is fetched from its low end and turned round) */
	const int32_t jl = L.rev ? (int32_t)(L.plen - L.qlen + dd) : (int32_t)(L.qlen - 1u - dd) - 15;
	const uint32_t ts = S.tosStride;
	uint32_t seg;
	if (jl >= 0) {
		const uint32_t w = (uint32_t)jl >> 4, sh = 2u * ((uint32_t)jl & 15u);
		const uint64_t two = (uint64_t)S.rl[w * ts] | ((uint64_t)S.rl[(w + 1u) * ts] << 32);
		seg = (uint32_t)(two >> sh);
	} else {
		seg = S.rl[0] << (2u * (uint32_t)(-jl));             /* -jl = 1..15 */
	}
	if (!L.rev) seg = bt_rev2(seg);
	if (!L.readFw) seg = ~seg;
	if (L.nmuts) {
		/* the seedling's substitutions (bt_apply_muts), by query index */
		const uint32_t r0 = L.qlen - 1u - L.mutpos0 - dd, r1 = L.qlen - 1u - L.mutpos1 - dd, r2 = L.qlen - 1u - L.mutpos2 - dd;
		if (r0 < 16u) seg = (seg & ~(3u << (2u * r0))) | ((uint32_t)L.mutnew0 << (2u * r0));
		if (L.nmuts > 1 && r1 < 16u) seg = (seg & ~(3u << (2u * r1))) | ((uint32_t)L.mutnew1 << (2u * r1));
		if (L.nmuts > 2 && r2 < 16u) seg = (seg & ~(3u << (2u * r2))) | ((uint32_t)L.mutnew2 << (2u * r2));
	}
	return seg;
}
/* How many of the depths [lo, hi) are alternatives as far as the qualities go (ham + penalty(q) <= qualThresh): what tells a
 * mapLFEx step from a mapLF1 step in a stretch that is skipped.  Four quality bytes per LDS word. */
BT_HD uint32_t bt_loc_count_alt(const BtLane& L, const BtScratch& S, uint32_t lo, uint32_t hi)
{
	if (hi <= lo) return 0;
	if (!L.considerQuals) return hi - lo;
	if (L.ham > L.qualThresh) return 0;
	const uint32_t P = L.qualThresh - L.ham;
	uint32_t qlim;
	if (!L.maq) qlim = P > 93u ? 93u : P;
	else qlim = P >= 30u ? 93u : P >= 20u ? 24u : P >= 10u ? 14u : 4u;
	const uint32_t thr = (qlim + 33u) * 0x01010101u | 0x80808080u;       /* a byte v <= qlim + 33  <=>  bit 7 of (0x80 + qlim + 33 - v) */
	const uint32_t jlo = L.rev ? L.plen - L.qlen + lo : L.qlen - hi, jhi = jlo + (hi - lo);
	const uint32_t ts = S.tosStride;
	uint32_t n = 0;
	BT_NOUNROLL
	for (uint32_t w = jlo >> 2; w <= (jhi - 1u) >> 2; w++) {
		uint32_t m = (thr - (S.rl[(BT_RL_SEQ_WORDS + w) * ts] & 0x7f7f7f7fu)) & 0x80808080u;
		if (w == jlo >> 2) m &= 0xffffffffu << (8u * (jlo & 3u));
		if (w == (jhi - 1u) >> 2) m &= 0xffffffffu >> (8u * (3u - ((jhi - 1u) & 3u)));
		n += (uint32_t)__builtin_popcount(m);
	}
	return n;
}
/* A window of the text in depth order: eight words (sixteen characters each) starting `sh` characters before depth w0; the
 * first depth at or after d, and below lim, at which the query differs from it (0xffffffff: none), and the text's base there. */
struct BtLocWin { uint32_t w0, w1, w2, w3, w4, w5, w6, w7; };          /* named words: registers are not indexed at run time */
BT_HD uint32_t bt_loc_win_word(const BtLocWin& t, uint32_t k)
{
	const uint32_t a = (k & 1u) ? t.w1 : t.w0, b = (k & 1u) ? t.w3 : t.w2, c = (k & 1u) ? t.w5 : t.w4, e = (k & 1u) ? t.w7 : t.w6;
	const uint32_t lo = (k & 2u) ? b : a, hi = (k & 2u) ? e : c;
	return k >= 8u ? 0u : ((k & 4u) ? hi : lo);
}
/* A window of the text in depth order -- words of sixteen characters, the first one starting `sh` characters before depth
 * w0 --: the first depth at or after d, and below lim, at which the query differs from it (0xffffffff: none), and the
 * text's base there.  Word by word from the shallow end, done at the first difference: a row that is not where the read
 * comes from differs within a base or two. */
BT_HD uint32_t bt_loc_first_mm(const BtLane& L, const BtScratch& S, const BtLocWin& t, uint32_t sh, uint32_t w0,
                               uint32_t d, uint32_t lim, uint32_t* tOut)
{
	uint32_t fm = 0xffffffffu, tf = 0;
	BT_NOUNROLL
	for (uint32_t k = (d - w0) >> 4; w0 + 16u * k < lim; k++) {
		const uint32_t wb = w0 + 16u * k;
		const uint32_t td = (uint32_t)((((uint64_t)bt_loc_win_word(t, k + 1u) << 32) | bt_loc_win_word(t, k)) >> (2u * sh));
		uint32_t x = bt_loc_qword(L, S, wb) ^ td;
		x = (x | (x >> 1)) & 0x55555555u;
		if (d > wb) x &= 0xffffffffu << (2u * (d - wb));
		if (lim < wb + 16u) x &= 0xffffffffu >> (2u * (wb + 16u - lim));
		if (x) { const uint32_t b = (uint32_t)__builtin_ctz(x); fm = wb + (b >> 1); tf = (td >> b) & 3u; break; }
	}
	*tOut = tf;
	return fm;
}

/* Save the current frame before a child is entered: its record in HBM (two 16-byte stores and a word) and the LDS
 * top-of-stack copy, from which a child that fails pops it back without a fetch. */
BT_HD void bt_frame_push(BtLane& L, const BtScratch& S, uint32_t mm)
{
	const uint32_t ts = S.tosStride;
	uint32_t w[BT_TOS_WORDS];
	w[FR_W0] = L.depth | (L.d << 11);
	w[FR_W1] = L.ham | (L.lowAltQual << 16);
	const uint32_t ccSave = S.noCC == 2u ? 0u : (uint32_t)L.ccValid;     /* (no room for the saved frame's candidate: it is fetched again) */
	w[FR_W2] = L.fu | (L.f1 << 11) | (L.elcint << 22) | (L.elignore << 24) | (L.candValid << 25) | (ccSave << 26);
	w[FR_W3] = L.f2 | (L.f3 << 11) | (L.cel1 << 22) | (L.cel2 << 26);
	w[FR_W4] = L.altNum | (L.eligibleNum << 12) | (L.low2 << 24);
	w[FR_W5] = L.cand | (L.dcf << 11) | (L.lmode << 22) | (L.lt << 23) | (L.lz << 25) | (L.el << 26);
	w[FR_W6] = L.pi | (L.pj << 11) | (L.pel << 13);
	w[FR_EBASE] = L.ebase;
	w[FR_ANCHOR] = (uint32_t)L.top;
#if BT_L2_TALLY
	const uint32_t l2w = L.cand2 | (L.num2 << 11) | (L.l2v << 23);
	w[BT_TOS_WORDS - 1] = l2w;
#else
	const uint32_t l2w = 0;
#endif
	uint32_t* fr = S.a->frames + ((uint64_t)S.slot * S.a->frCap + L.sd) * BT_FR_WORDS;
	BtU4 q0, q1, q2; q0.x = w[0]; q0.y = w[1]; q0.z = w[2]; q0.w = w[3]; q1.x = w[4]; q1.y = w[5]; q1.z = w[6]; q1.w = w[7];
	q2.x = w[FR_ANCHOR]; q2.y = mm; q2.z = l2w; q2.w = 0;          /* FR_ANCHOR, FR_MM (again: the caller wrote it), FR_L2 */
	bt_st4(fr, q0); bt_st4(fr + 4, q1); bt_st4(fr + 8, q2);
	BT_UNROLL
	for (uint32_t k = 0; k < BT_TOS_WORDS; k++) S.tosRec[k * ts] = w[k];
	if (ccSave) {
		BT_UNROLL
		for (uint32_t k = 0; k < BT_CC_WORDS; k++) S.tos[(BT_CC_WORDS + BT_TOS_WORDS + k) * ts] = S.tos[k * ts];
	}
	L.tosFrame = L.sd; L.tosValid = 1;
}

/* ---- the slow states: everything that is not "next query position" / "next SA-walk step" ---- */
template <bool RL>
BT_HD void bt_lane_slow(BtLane& L, const BtProgram& P, const BtHot& H, const BtWarm& W, const BtCold& C, const BtScratch& S,
                        const BtRes& res, BtReq& req, unsigned long long* CNT)
{
	const BtIndexDev* IX = C.ix;
	const BtBatchDev& B = C.ring[L.bid];
	/* The states are visited in an order that lets the usual chains finish in one sweep (child
	 * failed: FRAME_RETURN -> CHILD_RET -> BT_LOOP; phase change: FRAME_RETURN -> SEARCH_END ->
	 * PHASE_NEXT -> SEARCH_BEGIN).  Within a block `break` leaves the block; a block that sets `req`
	 * ends the lane's round: it leaves the lane in a state no LATER block of the sweep takes (ST_IS does not test req; the three
	 * blocks entered straight from the one before them use ST_IS_NOREQ), and the while condition ends the sweep. */
	while (BT_IS_SLOW(L.state) && req.kind == RQ_NONE) {
		BT_PROF_PASS();
		/* ---- ran off the 5' end of the query (:1086-1090) ------------------------------- */
		if (ST_IS(ST_FELL_OFF)) { BT_PROF_T0(t_fell_off); do {
			if (L.sd >= L.reportPartials) BT_GOTO_RA(L.sd, L.top, L.bot, L.ham, RC_FELL, L.lmode);
			else { L.ret = 0; L.state = ST_FRAME_RETURN; }
		} while (0); BT_PROF_ADD(PS_FELL_OFF, t_fell_off); }

		/* ---- reportAlignment / reportFullAlignment (:1455-1565) -------------------------- */
		if (ST_IS(ST_RA_BEGIN)) { BT_PROF_T0(t_ra_begin); do {
			if (L.reportPartials) {
				if (L.ra_sd > 0) bt_report_partial(L, S, L.ra_sd);
				L.ret = 0; L.state = ST_RA_END; break;
			}
			uint32_t stratum = 0;
			BT_NOUNROLL
			for (uint32_t i = 0; i < L.ra_sd; i++)                      /* calcStratum (:1164-1177) */
				if ((BT_FR_MM(i) & 0xffffu) >= (L.qlen - L.r3)) stratum++;
			stratum += L.nmuts;
			L.ra_stratum = stratum;
			L.ra_cost = (L.ra_cost | (stratum << 14)) & 0xffffu;
			if (L.ra_sd + L.nmuts == 0 && !L.reportExacts) { L.ret = 0; L.state = ST_RA_END; break; }
			{
				const bt_row spread = L.ra_bot - L.ra_top;
				bt_row r = bt_rnd_u32(L);
				if (C.ix[0].wide) {                                     /* nextU<TIndexOffU>() of the 64-bit build: random_source.h:56-62 */
					const uint64_t r64 = ((uint64_t)r << 32) | bt_rnd_u32(L);
					r = (bt_row)(r64 % spread);
				} else r %= spread;
				L.ra_r = L.ra_top + r;
				L.ra_i = 0;
			}
			L.state = ST_ROW_BEGIN;
		} while (0); BT_PROF_ADD(PS_RA_BEGIN, t_ra_begin); }

		if (ST_IS(ST_ROW_BEGIN)) { BT_PROF_T0(t_row_begin); do {
			const bt_row spread = L.ra_bot - L.ra_top;
			if (L.ra_i >= spread) { L.ret = 0; L.state = ST_RA_END; break; }
			bt_row ri = L.ra_r + L.ra_i;
			if (ri >= L.ra_bot) ri -= spread;
			L.crow = ri; L.cjumps = 0;
			L.state = (BT_LOC(RL) && L.ra_l) ? ST_RESOLVE_DONE : ST_CHASE_CHECK;       /* in locus mode "the row" is the anchor: nothing to walk */
		} while (0); BT_PROF_ADD(PS_ROW_BEGIN, t_row_begin); }

		/* ---- an SA walk reached a sampled row: offset -> (tidx,toff) -> sink (ebwt.h:2569-2746) ---- */
		if (ST_IS(ST_RESOLVE_DONE)) { BT_PROF_T0(t_resolve_done); do {
			const bt_row zOff = WSEL(zOff);
			bt_row off;
			if (!BT_WIDE && W.locOn) {
				/* no walk: the alignment was found in locus mode (its text offset is known: anchor - qlen), or the row's
				 * locus record has just arrived (word 0 = SA[row]).  The walk the reference does from here is tallied from
				 * the table of walk lengths */
				off = (RL && L.ra_l) ? L.crow - L.qlen : res.q[0].x;
				BT_COUNT_N(CN_TCHASE, BT_GP(const uint16_t, WSEL(walk))[off]);
			}
			else if (L.crow == zOff) off = L.cjumps;
			else off = bt_piece_row(res.q[0], (uint32_t)(L.crow >> WSEL(offRate))) + L.cjumps;
			BT_COUNT(CN_OFFS);
			/* joinedToTextOff (ebwt.h:2569-2629) */
			const bt_row* rstarts = IXSEL(rstarts);
			const uint32_t nFrag = IXSEL(nFrag), ixfw = L.mirror ? 0u : 1u;
			const bt_row len = WSEL(len);
			uint32_t lo = 0, hi = nFrag, tidx = 0, toff = 0, probes = 0;
			bool hit = false;
			BT_NOUNROLL
			for (;;) {
				const uint32_t elt = lo + ((hi - lo) >> 1);
				const bt_row lower = BT_GP(const bt_row, rstarts)[elt * 3u];
				const bt_row upper = (elt == nFrag - 1u) ? len : BT_GP(const bt_row, rstarts)[(elt + 1u) * 3u];
				probes++;
				if (lower <= off) {
					if (upper > off) {
						if (off + L.qlen <= upper) {
							bt_row fragoff = off - lower;
							if (!ixfw) { fragoff = (upper - lower) - fragoff - 1u; fragoff -= (L.qlen - 1u); }
							tidx = (uint32_t)BT_GP(const bt_row, rstarts)[elt * 3u + 1u];
							toff = (uint32_t)(fragoff + BT_GP(const bt_row, rstarts)[elt * 3u + 2u]);
							hit = true;
						}
						break;
					}
					lo = elt;
				} else hi = elt;
			}
			BT_COUNT_N(CN_RSTARTS, probes);
			if (hit && bt_report_hit(L, P, ixfw, S, B, tidx, toff)) { L.ret = 1; L.state = ST_RA_END; break; }
			L.ra_i++;
			L.state = ST_ROW_BEGIN;
		} while (0); BT_PROF_ADD(PS_RESOLVE_DONE, t_resolve_done); }

		if (ST_IS(ST_RA_END)) { BT_PROF_T0(t_ra_end); do {
			switch (L.ra_cont) {
			case RC_STEP:
				if (L.ret) { L.state = ST_FRAME_RETURN; break; }
				if (BT_LOC(RL) && L.lmode) L.bot = L.top; else L.top = L.bot;     /* keep looking (:730-735); in locus mode `top` is the anchor and stays */
				if (L.altNum > 0) L.state = ST_BT_LOOP;
				else { L.ret = 0; L.state = ST_FRAME_RETURN; }
				break;
			case RC_CHILD: L.state = ST_CHILD_RET; break;
			case RC_FELL:  L.state = ST_FRAME_RETURN; break;
			default:       L.state = ST_SEARCH_END; break;
			}
		} while (0); BT_PROF_ADD(PS_RA_END, t_ra_end); }

		/* ---- return from a frame: pop the parent's record (LDS copy, else one fetch) ---------- */
		if (ST_IS(ST_FRAME_RETURN) || ST_IS(ST_FRAME_FETCHED)) { BT_PROF_T0(t_frame_return); do {
			/* "child true => return true" (:972-974) all the way up: nothing of the frames in between is looked at again */
			if (L.ret && L.state == ST_FRAME_RETURN) { L.sd = 0; L.el = 0; L.state = ST_SEARCH_END; break; }
			/* the levels gone through without a frame (bt_loc_descend) return with the frame they started from */
			if (L.sd == L.el) { L.state = ST_SEARCH_END; break; }
			const uint32_t f = L.sd - L.el - 1u;
			uint32_t w[BT_TOS_WORDS];
			bool fromTos = false;
			if (L.state == ST_FRAME_FETCHED) {
				w[0] = res.q[0].x; w[1] = res.q[0].y; w[2] = res.q[0].z; w[3] = res.q[0].w;
				w[4] = res.q[1].x; w[5] = res.q[1].y; w[6] = res.q[1].z; w[7] = res.q[1].w;
				w[8] = res.q[2].x;
#if BT_L2_TALLY
				w[BT_TOS_WORDS - 1] = res.q[2].z;
#endif
			} else if (L.tosValid && L.tosFrame == f) {
				const uint32_t ts = S.tosStride;
				BT_UNROLL
				for (uint32_t k = 0; k < BT_TOS_WORDS; k++) w[k] = S.tosRec[k * ts];
				L.tosValid = 0;
				fromTos = true;
			} else {
				BT_REQ_FETCH(&FRW(f, 0), 3, nullptr);
				L.state = ST_FRAME_FETCHED;
				break;
			}
			L.sd = f;
			uint32_t v;
			v = w[FR_W0]; L.depth = v & 0x7ffu; L.d = (v >> 11) & 0x7ffu;
			v = w[FR_W1]; L.ham = v & 0xffffu; L.lowAltQual = (v >> 16) & 0xffu;
			v = w[FR_W2]; L.fu = v & 0x7ffu; L.f1 = (v >> 11) & 0x7ffu; L.elcint = (v >> 22) & 3u;
			L.elignore = (v >> 24) & 1u; L.candValid = (v >> 25) & 1u;
			L.ccValid = fromTos ? ((v >> 26) & 1u) : 0u;
			if (L.ccValid) {
				const uint32_t ts = S.tosStride;
				BT_UNROLL
				for (uint32_t k = 0; k < BT_CC_WORDS; k++) S.tos[k * ts] = S.tos[(BT_CC_WORDS + BT_TOS_WORDS + k) * ts];
			}
			v = w[FR_W3]; L.f2 = v & 0x7ffu; L.f3 = (v >> 11) & 0x7ffu; L.cel1 = (v >> 22) & 15u; L.cel2 = (v >> 26) & 15u;
			v = w[FR_W4]; L.altNum = v & 0xfffu; L.eligibleNum = (v >> 12) & 0xfffu; L.low2 = v >> 24;
#if BT_L2_TALLY
			v = w[BT_TOS_WORDS - 1]; L.cand2 = v & 0x7ffu; L.num2 = (v >> 11) & 0xfffu; L.l2v = (v >> 23) & 1u;
#endif
			v = w[FR_W5]; L.cand = v & 0x7ffu; L.dcf = (v >> 11) & 0x7ffu; L.lmode = (v >> 22) & 1u; L.lt = (v >> 23) & 3u; L.lz = (v >> 25) & 1u; L.el = v >> 26;
			v = w[FR_W6]; L.pi = v & 0x7ffu; L.pj = (v >> 11) & 3u; L.pel = (v >> 13) & 15u;
			L.ebase = w[FR_EBASE];
			if (L.lmode) { L.top = w[FR_ANCHOR]; L.bot = L.top; }      /* the anchor; a frame that has a child stands on an empty range */
			L.state = ST_CHILD_RET;
		} while (0); BT_PROF_ADD(PS_FRAME_RETURN, t_frame_return); }

		/* ---- a child frame (or a leaf report) came back (:972-1064) ---------------------- */
		if (ST_IS(ST_CHILD_RET)) { BT_PROF_T0(t_child_ret); do {
			if (L.ret) { L.state = ST_FRAME_RETURN; break; }
			if (L.bailed || (L.halfAndHalf && P.steps[L.step].maxBts > 0 && L.numBts >= P.steps[L.step].maxBts)) {
				L.bailed = 1; L.ret = 0; L.state = ST_FRAME_RETURN; break;
			}
			{
				const uint32_t e = bt_ent(L, L.pi);
				const uint32_t el = L.pel | (1u << L.pj);           /* the mask travelled with the frame record */
				META_MASK(e) = (uint8_t)el;
				L.pel = el;
				if (L.ccValid && L.pi == L.cand) S.tos[8u * S.tosStride] = (S.tos[8u * S.tosStride] & ~15u) | el;
				if (el == 15u) { L.candValid = 0; L.ccValid = 0; }      /* that position is exhausted: scan next time */
			}
			L.eligibleNum = L.eligibleNum - 1u;
			L.elignore = 1;
			L.altNum = L.altNum - 1u;
			if (L.altNum == 0) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
			if (L.eligibleNum == 0 && L.considerQuals) {
				/* re-scan the frame for the next-lowest-quality set of targets (:1004-1058), one
				 * fetched batch of records per round, deepest position first */
				BT_COUNT(CN_RESCAN);
				L.lowAltQual = 0xff; L.candValid = 0; L.ccValid = 0;
				const uint32_t kmin = L.depth > L.fu ? L.depth : L.fu;
#if BT_L2_TALLY
#if defined(BT_L2_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
				{ extern unsigned long long g_l2_hit, g_l2_miss, g_l2_none; if (L.d >= kmin) { if (L.l2v) g_l2_hit++; else g_l2_miss++; } else g_l2_none++; }
#endif
				if (L.d >= kmin && L.l2v) {
					/* the next level was tallied while the frame stepped forward (BtLane::low2): what the scan would find */
					L.lowAltQual = L.low2; L.eligibleNum = L.num2; L.cand = L.cand2; L.candValid = 1;
					const uint32_t ce = L.cel2;
					L.elcint = (ce & 1u) == 0 ? 0u : (ce & 2u) == 0 ? 1u : (ce & 4u) == 0 ? 2u : 3u;
					L.elignore = 0; L.l2v = 0;
					L.state = ST_BT_LOOP; break;
				}
#endif
				if (L.d >= kmin) { L.scanCb = bt_ent(L, L.d) >> 3; L.state = ST_RESCAN; break; }
			}
			L.state = ST_BT_LOOP;
		} while (0); BT_PROF_ADD(PS_CHILD_RET, t_child_ret); }

		if (ST_IS(ST_RESCAN) || ST_IS(ST_RESCAN_DONE)) { BT_PROF_T0(t_rescan); do {
			const uint32_t kmin = L.depth > L.fu ? L.depth : L.fu;
			const uint32_t e_lo = bt_ent(L, kmin), e_hi = bt_ent(L, L.d);
			const uint32_t c_lo = e_lo >> 3;
			if (L.state == ST_RESCAN) { bt_scan_request(L, S, c_lo, req); L.state = ST_RESCAN_DONE; break; }
			/* the batch (chunks lo..hi) arrived in res.q[0..]; same walk as the reference: deepest
			 * record first, restart the tallies whenever a strictly lower quality shows up.  The
			 * affordability test ham + penalty(q) <= qualThresh is folded into one bound on q, and
			 * the running tallies stay in plain registers until the batch is done. */
			const uint32_t hi = L.scanCb, lo = hi >= c_lo + 3u ? hi - 3u : c_lo;
			uint32_t qlim;                                     /* records with q > qlim cannot be afforded */
			{
				const uint32_t P = L.qualThresh >= L.ham ? L.qualThresh - L.ham : 0xffffffffu;
				if (P == 0xffffffffu) qlim = 0xffffffffu;         /* nothing affordable: marks "none" below */
				else if (!L.maq) qlim = P > 0xffu ? 0xffu : P;
				else qlim = P >= 30u ? 0xffu : P >= 20u ? 24u : P >= 10u ? 14u : 4u;
			}
			if (qlim != 0xffffffffu) {
				BtScanAcc a; a.low = L.lowAltQual; a.num = L.eligibleNum; a.cnd = 0xffffffffu; a.cel = 0;
				const uint32_t span = e_hi - e_lo, t0 = lo * 8u - e_lo, np = hi - lo;     /* t0 may be "negative": wraps above span */
				BT_NOUNROLL
				for (int t = 3; t >= 0; t--) {                    /* one copy of the eight-record walk: the kernel is short of registers */
					if ((uint32_t)t > np) continue;
					BtU4 pc;
					pc.x = bt_sel4((uint32_t)t, res.q[0].x, res.q[1].x, res.q[2].x, res.q[3].x); pc.y = bt_sel4((uint32_t)t, res.q[0].y, res.q[1].y, res.q[2].y, res.q[3].y);
					pc.z = bt_sel4((uint32_t)t, res.q[0].z, res.q[1].z, res.q[2].z, res.q[3].z); pc.w = bt_sel4((uint32_t)t, res.q[0].w, res.q[1].w, res.q[2].w, res.q[3].w);
					bt_rescan_piece(pc, t0 + 8u * (uint32_t)t, span, qlim, a);
				}
				L.lowAltQual = a.low; L.eligibleNum = a.num;
				if (a.cnd != 0xffffffffu) {
					L.cand = bt_ent_pos(L, kmin, a.cnd); L.candValid = 1; L.ccValid = 0;
					L.elcint = (a.cel & 1u) == 0 ? 0u : (a.cel & 2u) == 0 ? 1u : (a.cel & 4u) == 0 ? 2u : 3u;
					L.elignore = 0;
				}
			}
			if (lo > c_lo) { L.scanCb = lo - 1u; L.state = ST_RESCAN; break; }
			L.state = ST_BT_LOOP;
		} while (0); BT_PROF_ADD(PS_RESCAN, t_rescan); }

		/* ---- backtrack() exit (:333-353, 303-324) + the seedling-extension loop ---------- */
		if (ST_IS(ST_SEARCH_END)) { BT_PROF_T0(t_search_end); do {
			L.numBts = 0;
			if (L.kind == BT_KIND_EXTEND) {
				/* search_seeded_phase3.c:9-59 / phase4.c:9-55: for each seedling, setMuts +
				 * backtrack(oldQuals); the RNG runs on across seedlings */
				if (!L.palIdxBefore && L.ret) { bt_lane_finish(L, B); break; }
				if (L.palIdxBefore) { L.palIdxBefore = 0; L.palIdx = 0; } else L.palIdx = L.palIdx + 1u;
				L.nmuts = 0;
				if (L.palIdx >= L.npals) { L.state = ST_PHASE_NEXT; break; }
				/* PartialAlignmentManager::toMutsString (ebwt_search_util.h:310-362) */
				const uint64_t pal = PALS(L.palIdx);
				const uint32_t p0 = (uint32_t)(pal & 0xffffu), p1 = (uint32_t)((pal >> 16) & 0xffffu), p2 = (uint32_t)((pal >> 32) & 0xffffu);
				uint32_t oldQuals = 0, nm = 1;
				const uint32_t t0 = L.plen - 1u - p0;
				oldQuals = (oldQuals + bt_mm_penalty(L.maq, bt_qual<RL>(L, H, S, t0))) & 0xffu;
				L.mutpos0 = t0; L.mutnew0 = (uint32_t)((pal >> 48) & 3u);
				if (p1 != 0xffffu) {
					const uint32_t t1 = L.plen - 1u - p1;
					oldQuals = (oldQuals + bt_mm_penalty(L.maq, bt_qual<RL>(L, H, S, t1))) & 0xffu;
					L.mutpos1 = t1; L.mutnew1 = (uint32_t)((pal >> 50) & 3u); nm = 2;
					if (p2 != 0xffffu) {
						const uint32_t t2 = L.plen - 1u - p2;
						oldQuals = (oldQuals + bt_mm_penalty(L.maq, bt_qual<RL>(L, H, S, t2))) & 0xffu;
						L.mutpos2 = t2; L.mutnew2 = (uint32_t)((pal >> 52) & 3u); nm = 3;
					}
				}
				L.nmuts = nm;
				L.iham = oldQuals;
				L.state = ST_SEARCH_BEGIN;
				break;
			}
			if (L.kind == BT_KIND_GEN) { L.state = ST_PHASE_NEXT; break; }
			if (L.ret) { bt_lane_finish(L, B); break; }
			L.state = ST_PHASE_NEXT;
		} while (0); BT_PROF_ADD(PS_SEARCH_END, t_search_end); }

		/* ---- phase script ------------------------------------------------------------- */
		if (ST_IS(ST_PHASE_NEXT)) { BT_PROF_T0(t_phase_next); do {
			L.step = L.step + 1u;        /* 5-bit wrap: 31 -> 0 */
			if ((int32_t)L.step >= P.nsteps || (L.status & BT_STF_OVERFLOW)) { bt_lane_finish(L, B); break; }
			const BtStep& st = P.steps[L.step];
			/* setQuery + setOffs + ctor flags */
			L.mirror = st.mirror; L.readFw = st.readFw; L.rev = ((st.mirror != 0) == (st.readFw != 0)) ? 1u : 0u;
			L.kind = st.kind; L.reportExacts = st.reportExacts; L.considerQuals = st.considerQuals;
			L.halfAndHalf = st.halfAndHalf; L.maq = st.maq; L.reportPartials = st.reportPartials;
			L.qualThresh = st.qualThresh;
			const uint32_t plen = L.plen, qs = plen < P.seedLen ? plen : P.seedLen;
			L.d5 = bt_off_code(plen, qs, st.oc[0]); L.d3 = bt_off_code(plen, qs, st.oc[1]);
			L.unrev = bt_off_code(plen, qs, st.oc[2]); L.r1 = bt_off_code(plen, qs, st.oc[3]);
			L.r2 = bt_off_code(plen, qs, st.oc[4]); L.r3 = bt_off_code(plen, qs, st.oc[5]);
			L.qlen = (st.kind == BT_KIND_GEN) ? qs : plen;               /* setQlen(seed) */
			L.rnd = L.seed; L.numBts = 0; L.nmuts = 0; L.iham = 0;
			if (st.kind == BT_KIND_GEN) L.npals = 0;
			if (st.kind == BT_KIND_EXTEND) {
				if (L.npals == 0) { L.state = ST_PHASE_NEXT; break; }
				L.palIdxBefore = 1; L.palIdx = 0; L.ret = 0;
				L.state = ST_SEARCH_END;
				break;
			}
			L.state = ST_SEARCH_BEGIN;
		} while (0); BT_PROF_ADD(PS_PHASE_NEXT, t_phase_next); }

		/* ---- backtrack() entry: tallyNs + ftab jump (:237-297, 1308-1362) -------------- */
		if (ST_IS(ST_SEARCH_BEGIN)) { BT_PROF_T0(t_search_begin); do {
			L.bailed = 0; L.sd = 0; L.lmode = 0; L.dcf = 0; L.el = 0;
			uint32_t nsInFtab = 0;
			const uint32_t ftabChars = WSEL(ftabChars);
			if (L.hasN) {
				/* tallyNs (:1308-1341); reads without any N (nearly all) skip both walks */
				uint32_t nsInSeed = 0; bool ok = true;
				BT_NOUNROLL
				for (uint32_t i = 0; i < L.r3 && ok; i++) {
					if (bt_qry<RL>(L, H, S, L.qlen - i - 1u) == 4u) {
						nsInSeed++;
						if (nsInSeed == 1) { if (i < L.unrev) ok = false; }
						else if (nsInSeed == 2) { if (i < L.r1) ok = false; }
						else if (nsInSeed == 3) { if (i < L.r2) ok = false; }
						else ok = false;
					}
				}
				if (!ok) { L.ret = 0; L.state = ST_SEARCH_END; break; }
				BT_NOUNROLL
				for (uint32_t i = 0; i < ftabChars && i < L.qlen; i++)
					if (bt_qry<RL>(L, H, S, L.qlen - i - 1u) == 4u) nsInFtab++;
			}
			L.nsFtab0 = nsInFtab > 0 ? 1u : 0u;
			const uint32_t m = L.unrev < L.qlen ? L.unrev : L.qlen;
			L.fu = L.unrev; L.f1 = L.r1; L.f2 = L.r2; L.f3 = L.r3; L.ham = L.iham; L.ebase = 0;
			const uint32_t jumpChars = BT_WIDE ? 0u : WSEL(jumpChars);
			if (!BT_WIDE && RL && jumpChars > ftabChars && !L.hasN && m >= jumpChars && L.qlen > jumpChars) {
				/* the first jumpChars positions are ones the search may not revisit: what ftab + the steps behind it come
				 * to is in the jump table (bt_rank.h) */
				uint32_t off = 0;
				BT_NOUNROLL
				for (uint32_t t = 0; t < jumpChars; t++) off |= bt_qry<RL>(L, H, S, L.qlen - 1u - t) << (2u * t);
				L.ra_r = off;           /* parked until the entry arrives */
				BT_REQ_FETCH(WSEL(jump) + (size_t)(off >> 1) * 4u, 1, WSEL(jumpMeta) + (off & ~7u));
				L.state = ST_JUMP_DONE;
			} else if (RL && nsInFtab == 0 && m >= ftabChars) {
				/* calcFtabOff (:1348-1362) straight from the LDS copy of the read */
				uint32_t ftabOff = 0;
				BT_NOUNROLL
				for (uint32_t t = 0; t < ftabChars; t++) ftabOff |= bt_qry<RL>(L, H, S, L.qlen - 1u - t) << (2u * t);
				const bt_row* ftab = WSEL(ftab);
				L.ra_r = ftabOff;           /* parked until the table entry arrives */
				BT_REQ_FETCH(ftab + (ftabOff & ~(BT_PIECE_ROWS - 1u)), 1, ftab + ((ftabOff + 1u) & ~(BT_PIECE_ROWS - 1u)));
				L.state = ST_FTAB_DONE;
			} else if (nsInFtab == 0 && m >= ftabChars) {
				/* calcFtabOff (:1348-1362) needs the last ftabChars characters of the query: fetch the
				 * (at most two) 16-byte chunks of the read that hold them */
				const uint32_t i0 = L.qlen - ftabChars, i1 = L.qlen - 1u;
				const uint32_t j0 = L.rev ? (L.plen - 1u - i1) : i0, j1 = L.rev ? (L.plen - 1u - i0) : i1;
				const uint8_t* base = H.seq + L.roff;
				BT_REQ_FETCH(base + (uint64_t)(j0 >> 4) * 16u, 1, base + (uint64_t)(j1 >> 4) * 16u);
				L.state = ST_FTABSEQ_DONE;
			} else {
				L.depth = 0; L.top = 0; L.bot = 0; L.state = ST_FRAME_ENTER;
			}
		} while (0); BT_PROF_ADD(PS_SEARCH_BEGIN, t_search_begin); }

		if (ST_IS_NOREQ(ST_FTABSEQ_DONE)) { BT_PROF_T0(t_ftabseq_done); do {
			const uint32_t ftabChars = WSEL(ftabChars);
			const uint32_t i0 = L.qlen - ftabChars, i1 = L.qlen - 1u;
			const uint32_t j0 = L.rev ? (L.plen - 1u - i1) : i0;
			uint32_t ftabOff = 0;
			BT_NOUNROLL
			for (uint32_t t = 0; t < ftabChars; t++) {
				const uint32_t i = L.qlen - 1u - t;
				const uint32_t j = L.rev ? (L.plen - 1u - i) : i;
				uint32_t c = ((j >> 4) == (j0 >> 4)) ? bt_u4_byte(res.q[0], j & 15u) : bt_u4_byte(res.x, j & 15u);
				if (!L.readFw && c < 4u) c ^= 3u;
				c = bt_apply_muts(L, i, c);
				ftabOff |= c << (2u * t);
			}
			const bt_row* ftab = WSEL(ftab);
			L.ra_r = ftabOff;           /* parked until the table entry arrives */
			BT_REQ_FETCH(ftab + (ftabOff & ~(BT_PIECE_ROWS - 1u)), 1, ftab + ((ftabOff + 1u) & ~(BT_PIECE_ROWS - 1u)));
			L.state = ST_FTAB_DONE;
		} while (0); BT_PROF_ADD(PS_FTABSEQ_DONE, t_ftabseq_done); }

		if (ST_IS_NOREQ(ST_FTAB_DONE)) { BT_PROF_T0(t_ftab_done); do {
			const uint32_t ftabChars = WSEL(ftabChars);
			const bt_row len = BT_WROWLIM();
			const uint32_t ftabOff = (uint32_t)L.ra_r;
			bt_row top = bt_piece_row(res.q[0], ftabOff);
			bt_row bot = (((ftabOff + 1u) & ~(BT_PIECE_ROWS - 1u)) == (ftabOff & ~(BT_PIECE_ROWS - 1u))) ? bt_piece_row(res.q[0], ftabOff + 1u)
			                                                                                          : bt_piece_row(res.x, ftabOff + 1u);
			if (top > len) { const bt_row* eftab = IXSEL(eftab); top = BT_GP(const bt_row, eftab)[(top ^ BT_OFF_MASK) * 2u + 1u]; }
			if (bot > len) { const bt_row* eftab = IXSEL(eftab); bot = BT_GP(const bt_row, eftab)[(bot ^ BT_OFF_MASK) * 2u]; }
			BT_COUNT(CN_FTAB);
			if (L.qlen == ftabChars && bot > top) {
				if (L.reportPartials > 0) { L.depth = 0; L.top = 0; L.bot = 0; L.state = ST_FRAME_ENTER; }
				else { L.top = top; L.bot = bot; BT_GOTO_RA(0, top, bot, L.iham, RC_ENTRY, 0); }
			} else if (bot > top) {
				L.depth = ftabChars; L.top = top; L.bot = bot; L.state = ST_FRAME_ENTER;
			} else { L.ret = 0; L.state = ST_SEARCH_END; }
		} while (0); BT_PROF_ADD(PS_FTAB_DONE, t_ftab_done); }

		if (!BT_WIDE && ST_IS_NOREQ(ST_JUMP_DONE)) { do {
			const uint32_t off = (uint32_t)L.ra_r;
			const uint32_t top = bt_u4_word(res.q[0], (off & 1u) * 2u), bot = bt_u4_word(res.q[0], (off & 1u) * 2u + 1u);
			const uint32_t meta = bt_u4_meta(res.x, off & 7u);
			const uint32_t nOps = meta & 7u, nMulti = (meta >> 3) & 7u, nSame = (meta >> 6) & 7u;
			/* what the reference goes through for this: the ftab look-up and the steps behind it */
			BT_COUNT(CN_FTAB); BT_COUNT(CN_JUMPS);
			BT_COUNT_N(CN_JLF2, nMulti); BT_COUNT_N(CN_JLF1, nOps - nMulti); BT_COUNT_N(CN_JSAME, nSame);
			if (bot > top) { L.depth = WSEL(jumpChars); L.top = top; L.bot = bot; L.state = ST_FRAME_ENTER; }
			else {
				/* the range ran empty on the way: the reference entered its frame behind ftab (unless ftab's own range was
				 * empty: no steps) and the frame failed without anything to go back to */
				if (nOps > 0) BT_COUNT(CN_FRAMES);
				L.ret = 0; L.state = ST_SEARCH_END;
			}
		} while (0); }

		/* ---- choose a backtrack target and descend (:743-971) --------------------------- */
		if (ST_IS(ST_BT_LOOP)) { BT_PROF_T0(t_bt_loop); do {
			if (!L.candValid) {
				/* the deepest position that still has a target of the eligible quality (the
				 * `for(; i >= depth; i--)` walk of :767-812), batch by batch */
				BT_COUNT(CN_CANDSCAN);
				L.scanCb = bt_ent(L, L.d) >> 3;
				L.state = ST_CANDSCAN;
				break;
			}
			L.state = ST_BT_PICK;
			if (L.ccValid) break;                        /* its ranges are in LDS: pick right away */
			if (L.lmode && L.cand >= L.dcf) break;       /* the frame's position in the text: one alternative, the text's base (lt) */
			/* fetch the target position's four (top,bot) ranges and its (mask,quality) record */
			const uint32_t e = bt_ent(L, L.cand);
			BT_REQ_FETCH(&PT(e, 0), BT_ENT_PIECES, &META(e & ~7u));
		} while (0); BT_PROF_ADD(PS_BT_LOOP, t_bt_loop); }

		if (ST_IS(ST_CANDSCAN) || ST_IS(ST_CANDSCAN_DONE)) { BT_PROF_T0(t_candscan); do {
			const uint32_t e_lo = L.ebase, e_hi = bt_ent(L, L.d);
			const uint32_t c_lo = e_lo >> 3;
			if (L.state == ST_CANDSCAN) { bt_scan_request(L, S, c_lo, req); L.state = ST_CANDSCAN_DONE; break; }
			const uint32_t hi = L.scanCb, lo = hi >= c_lo + 3u ? hi - 3u : c_lo;
			uint32_t cnd = 0xffffffffu;
			{
				const uint32_t span = e_hi - e_lo, t0 = lo * 8u - e_lo, np = hi - lo;
				const bool anyq = !L.considerQuals;
				BT_NOUNROLL
				for (uint32_t t = 0; t <= np; t++) {
					BtU4 pc;
					pc.x = bt_sel4(t, res.q[0].x, res.q[1].x, res.q[2].x, res.q[3].x); pc.y = bt_sel4(t, res.q[0].y, res.q[1].y, res.q[2].y, res.q[3].y);
					pc.z = bt_sel4(t, res.q[0].z, res.q[1].z, res.q[2].z, res.q[3].z); pc.w = bt_sel4(t, res.q[0].w, res.q[1].w, res.q[2].w, res.q[3].w);
					bt_candscan_piece(pc, t0 + 8u * t, span, L.lowAltQual, anyq, cnd);
				}
			}
			const bool found = cnd != 0xffffffffu;
			if (found) { L.cand = bt_ent_pos(L, L.depth, cnd); L.candValid = 1; L.ccValid = 0; }
			if (found) { L.state = ST_BT_LOOP; break; }
			if (lo > c_lo) { L.scanCb = lo - 1u; L.state = ST_CANDSCAN; break; }
			L.state = ST_ABORT;                                   /* cannot happen: altNum > 0 */
		} while (0); BT_PROF_ADD(PS_CANDSCAN, t_candscan); }

		if (ST_IS_NOREQ(ST_BT_PICK)) { BT_PROF_T0(t_bt_pick); do {
			const uint32_t i = L.cand;
			const uint32_t e = bt_ent(L, i);
			const uint32_t ts = S.tosStride;
			uint32_t mv;
			bt_row tp[4], bp[4];
			const bool locTarget = L.lmode && i >= L.dcf;
			if (locTarget && !L.ccValid) {
				/* a position decided by the text has one alternative, the text's base there, on the one row that goes with it;
				 * it is picked at most once (afterwards its mask is full), so its mask is the fresh one */
				BT_UNROLL
				for (uint32_t k = 0; k < 4u; k++) { tp[k] = 0; bp[k] = (k == L.lt) ? 1u : 0u; }
				mv = (15u ^ (1u << L.lt)) | (bt_qual<RL>(L, H, S, L.qlen - i - 1u) << 8);
			} else if (L.ccValid) {
				BT_UNROLL
				for (uint32_t k = 0; k < 4u; k++) { tp[k] = S.tos[k * ts]; bp[k] = S.tos[(4u + k) * ts]; }
				mv = S.tos[8u * ts];
			} else {
				bt_res_quartets(res, tp, bp);
				mv = bt_u4_meta(res.x, e & 7u);
				if (!BT_WIDE && S.noCC != 1u) {
					BT_UNROLL
					for (uint32_t k = 0; k < 4u; k++) { S.tos[k * ts] = (uint32_t)tp[k]; S.tos[(4u + k) * ts] = (uint32_t)bp[k]; }
					S.tos[8u * ts] = mv;
					L.ccValid = 1;
				}
			}
			const uint32_t el = mv & 15u, qi = mv >> 8;
			const bt_row sp[4] = {bp[0] - tp[0], bp[1] - tp[1], bp[2] - tp[2], bp[3] - tp[3]};
			uint32_t j = 0;
			if (L.eligibleNum > 1 || L.elignore) {
				bt_row posSz = 0;
				BT_UNROLL
				for (uint32_t l = 0; l < 4u; l++) if ((el & (1u << l)) == 0) posSz += sp[l];
				if (posSz == 0) { L.state = ST_ABORT; break; }
				uint32_t r = (uint32_t)(bt_rnd_u32(L) % posSz);      /* (:788-795: a 32-bit draw and 32-bit spreads in the 64-bit build too) */
				bool found = false;
				BT_UNROLL
				for (uint32_t l = 0; l < 4u; l++) {
					if (!found && (el & (1u << l)) == 0) {
						if (r < (uint32_t)sp[l]) { j = l; found = true; }
						else r -= (uint32_t)sp[l];
					}
				}
			} else {
				j = L.elcint;                                     /* the only eligible target: no draw (:820-834) */
			}
			/* a child of a position in the text stands on the same anchor, one position further */
			const bt_row bttop = locTarget ? L.top : bt_selr4(j, tp[0], tp[1], tp[2], tp[3]);
			const bt_row btbot = bttop + bt_selr4(j, sp[0], sp[1], sp[2], sp[3]);
			const uint32_t btham = L.ham + bt_mm_penalty(L.maq, qi);
			const uint32_t btcint = j;
			L.pel = el;
			const uint32_t icur = L.qlen - i - 1u;
			uint32_t nu = L.fu, n1 = L.f1, n2 = L.f2, n3 = L.f3;
			if (i < L.f1)      { nu = L.f1; n1 = L.f2; n2 = L.f3; }
			else if (i < L.f2) { n1 = L.f2; n2 = L.f3; }
			else if (i < L.f3) { n2 = L.f3; }
			FRW(L.sd, FR_MM) = icur | (btcint << 16);
			L.pi = i; L.pj = j; L.btham = btham;
			if (i + 1u == L.qlen) {
				BT_GOTO_RA(L.sd + 1u, bttop, btbot, btham, RC_CHILD, locTarget);
				break;
			}
			uint32_t newDepth = i + 1u;
			bt_row ntop = bttop, nbot = btbot;
			bool childLoc = locTarget;
			const bool rootNoFtab = (L.sd == 0) && L.nsFtab0;
			const uint32_t ftabChars = WSEL(ftabChars);
			if (L.halfAndHalf && !rootNoFtab && L.r2 == L.r3 && i + 1u < ftabChars && ftabChars <= L.d5) {
				/* re-jump through the ftab with the substituted character (:908-952); rare, synchronous */
				uint32_t ftabOff = 0;
				BT_NOUNROLL
				for (uint32_t jj = 0; jj < ftabChars; jj++) {
					uint32_t c = bt_qry<RL>(L, H, S, L.qlen - 1u - jj);
					if (L.qlen - 1u - jj == icur) c = btcint;
					ftabOff |= c << (2u * jj);
				}
				const bt_row* ftab = WSEL(ftab); const bt_row* eftab = IXSEL(eftab); const bt_row len = BT_WROWLIM();
				ntop = BT_GP(const bt_row, ftab)[ftabOff]; nbot = BT_GP(const bt_row, ftab)[ftabOff + 1u];
				if (ntop > len) ntop = BT_GP(const bt_row, eftab)[(ntop ^ BT_OFF_MASK) * 2u + 1u];
				if (nbot > len) nbot = BT_GP(const bt_row, eftab)[(nbot ^ BT_OFF_MASK) * 2u];
				BT_COUNT(CN_FTAB);
				if (ntop == nbot) { L.ret = 0; L.state = ST_CHILD_RET; break; }
				newDepth = ftabChars;
				childLoc = false;                             /* rows again */
			}
			/* push: save the parent (HBM record + LDS top-of-stack copy), enter the child */
			if (L.sd + 1u >= S.a->frCap) { L.state = ST_ABORT; break; }
			bt_frame_push(L, S, icur | (btcint << 16));
			L.ebase = bt_ent(L, L.d) + 1u;
			L.lmode = childLoc ? 1u : 0u; L.dcf = childLoc ? newDepth : 0u; L.el = 0;
			L.sd = L.sd + 1u; L.depth = newDepth; L.top = ntop; L.bot = nbot; L.ham = btham;
			L.fu = nu; L.f1 = n1; L.f2 = n2; L.f3 = n3;
			L.state = ST_FRAME_ENTER;
		} while (0); BT_PROF_ADD(PS_BT_PICK, t_bt_pick); }

		/* ---- frame prologue (:363-455) -------------------------------------------------- */
		if (ST_IS(ST_FRAME_ENTER)) { BT_PROF_T0(t_frame_enter); do {
			BT_COUNT(CN_FRAMES);
			if (L.halfAndHalf) {
				const uint32_t maxBts = P.steps[L.step].maxBts;
				if (maxBts > 0 && L.numBts == maxBts) { L.bailed = 1; L.ret = 0; L.state = ST_FRAME_RETURN; break; }
				L.numBts++;
			}
			L.altNum = 0; L.eligibleNum = 0;
			L.elcint = 0; L.elignore = 1;
			L.lowAltQual = 0xff; L.candValid = 0; L.cand = 0; L.ccValid = 0; BT_L2_RESET(L);
			L.d = L.depth;
			L.state = ST_STEP_BEGIN;
		} while (0); BT_PROF_ADD(PS_FRAME_ENTER, t_frame_enter); }

		if (ST_IS(ST_ABORT)) { L.status = L.status | BT_STF_OVERFLOW; bt_lane_finish(L, B); }
	}
}

/* A frame in locus mode has just met the text's other base at its last position, as an alternative of the eligible
 * quality.  That position is the deepest of the frame, so it is the target the reference takes next (:767-834) -- with a
 * draw that decides nothing (r % 1) unless it is the single eligible target, in which case there is none -- and its child
 * stands on the same anchor, one position further.  Taking it needs nothing from memory, so it is done here, where the
 * step was decided, and not by the sweep's target choice:
 *   - a frame that has stood on the text from its first position (dcf == depth) has this ONE alternative in all; its only
 *     alternative spent, it returns what the child returns.  Such a frame needs no record: the lane becomes the child, and
 *     whatever comes back goes to the frame the chain of such levels started from (L.el counts them; ST_FRAME_RETURN skips
 *     them);
 *   - a frame with positions in row space as well is saved like any frame that gets a child (bt_frame_push).
 * false: not done here (the half-and-half re-jump through the ftab, the frame arena full) -- the caller goes the general way. */
template <bool RL>
BT_HD bool bt_loc_descend(BtLane& L, const BtProgram& P, const BtWarm& W, const BtScratch& S, unsigned long long* CNT)
{
	const uint32_t i = L.d, j = L.lt, icur = L.qlen - i - 1u;
	const bool pure = L.dcf == L.depth;
	if ((pure && L.el == 63u) || L.sd + 1u >= S.a->frCap) return false;
	const bool rootNoFtab = (L.sd == 0) && L.nsFtab0;
	if (L.halfAndHalf && !rootNoFtab && L.r2 == L.r3 && i + 1u < WSEL(ftabChars) && WSEL(ftabChars) <= L.d5) return false;
	if (L.eligibleNum > 1 || L.elignore) (void)bt_rnd_u32(L);        /* r % 1: the draw is made, its value decides nothing */
	const uint32_t btham = L.ham + bt_mm_penalty(L.maq, L.q);
	uint32_t nu = L.fu, n1 = L.f1, n2 = L.f2, n3 = L.f3;
	if (i < L.f1)      { nu = L.f1; n1 = L.f2; n2 = L.f3; }
	else if (i < L.f2) { n1 = L.f2; n2 = L.f3; }
	else if (i < L.f3) { n2 = L.f3; }
	FRW(L.sd, FR_MM) = icur | (j << 16);
	L.pel = 15u ^ (1u << j); L.pi = i; L.pj = j;
	if (i + 1u == L.qlen) {
		/* the substitution is the read's last position: reported as the child's alignment.  A frame with nothing else to
		 * try then returns what the report returns (RC_FELL: straight to ST_FRAME_RETURN) */
		BT_GOTO_RA(L.sd + 1u, L.top, L.top + 1u, btham, pure ? RC_FELL : RC_CHILD, 1);
		return true;
	}
	if (pure) L.el = L.el + 1u;
	else { bt_frame_push(L, S, icur | (j << 16)); L.el = 0; }
	/* the child's frame, entered (:363-455) */
	L.ebase = bt_ent(L, L.d) + 1u;
	L.sd = L.sd + 1u;
	L.depth = i + 1u; L.dcf = i + 1u; L.bot = L.top + 1u; L.ham = btham;
	L.fu = nu; L.f1 = n1; L.f2 = n2; L.f3 = n3;
	BT_COUNT(CN_FRAMES);
	if (L.halfAndHalf) {
		const uint32_t maxBts = P.steps[L.step].maxBts;
		if (maxBts > 0 && L.numBts == maxBts) { L.bailed = 1; L.ret = 0; L.state = ST_FRAME_RETURN; return true; }
		L.numBts++;
	}
	L.altNum = 0; L.eligibleNum = 0;
	L.elcint = 0; L.elignore = 1;
	L.lowAltQual = 0xff; L.candValid = 0; L.cand = 0; L.ccValid = 0; BT_L2_RESET(L);
	L.d = L.depth;
	L.state = ST_STEP_BEGIN;
	return true;
}

/*
 * Advance one lane until it has a memory request for this round (req.kind != RQ_NONE) or has
 * finished its read (state ST_IDLE).  `res` is the answer to the lane's previous request.
 */
template <bool RL>
BT_HD void bt_lane_run(BtLane& L, const BtProgram& P, const BtHot& H, const BtWarm& W, const BtCold& C, const BtScratch& S,
                       const BtRes& res, BtReq& req, unsigned long long* CNT)
{
	req.kind = RQ_NONE; req.n = 0; req.a = 0; req.x = 0; req.wchunk = 0xffffu;        /* (tally: the caller's to clear and to read) */
	/* locus mode: the window of text this call's answer holds (0 none, 1 a locus record's 48 characters, 2 pieces of the
	 * reversed text), the depth it starts at and the anchor it belongs to; whether the step just decided was a mismatch */
	uint32_t wkind = 0, wd0 = 0, wanchor = 0;
	bool locMiss = false;
	/* ONE pass per call -- arrivals, the step they decide, the slow-state sweep, the next request -- and where a lane could go
	 * on without memory (a position that needs no rank, the read's end, a check that fails) it waits for the next round's
	 * pass instead of going round again: a wavefront pays for every trip round this code with all its lanes, the trips it
	 * makes in a round are its slowest lane's, and a lane that waits a round costs nobody anything (scripts/pass_model.py:
	 * 1.3 trips per wavefront and round in row space, 3 in locus mode, for 1.01-1.07 per lane). */
	{
		BT_PROF_TICK(PS_RUN_ITERS);
		BT_PROF_T0(t_resume);
		if (BT_LOC(RL) && (L.state == ST_LOC_REC || L.state == ST_LOC_TXT)) {
			if (L.state == ST_LOC_REC) {
				/* the row's locus record: from here on the frame stands on a place in the text, not on a row */
				L.lmode = 1; L.dcf = L.d;
				L.top = res.q[0].x + L.d; L.bot = L.top + 1u;
				wkind = 1;
			} else wkind = 2;
			wd0 = L.d; wanchor = (uint32_t)L.top;
			L.state = ST_STEP_BEGIN;
		}
		if (BT_LOC(RL) && L.state == ST_STEP_BEGIN && L.lmode) {
			/* ---- locus mode: the next event of the frame (see "locus mode: the pieces") ------------------------ */
			BT_PROF_T0(t_locus);
			BT_PROF_TICK(PS_LOCUS_PASSES);
			const uint32_t d = L.d;
			if (d >= L.qlen) L.state = ST_FELL_OFF;
			else if (L.halfAndHalf && !bt_hh_check_top(L, S, d)) { L.ret = 0; L.state = ST_FRAME_RETURN; }
			else if (bt_ent(L, d) >= S.a->entCap) L.state = ST_ABORT;
			else {
				const uint32_t anchor = (uint32_t)L.top, len = (uint32_t)WSEL(len);
				const uint32_t y0 = len - anchor + wd0;                 /* first character of a kind-2 window in the reversed text */
				const uint32_t sh = wkind == 2 ? (y0 & 15u) : 0u;
				const uint32_t ncov = wkind == 2 ? 128u - sh : BT_LOC_CTX;
				if (wkind == 0 || wanchor != anchor || d < wd0 || d >= wd0 + ncov) {
					/* no text at hand for this depth: fetch what is left of the read's length (two 16-byte pieces hold 128
					 * characters; the address is a word's, not a piece's) */
					const uint32_t y = len - anchor + d;
					BT_REQ_FETCH(WSEL(rtxt) + (y >> 4), ((y & 15u) + (L.qlen - d) + 63u) >> 6, nullptr);
					L.state = ST_LOC_TXT;
					BT_COUNT_HOST(CN_FETCH);
					return;
				}
				const bool rec = wkind == 1;
				BtLocWin wt;
				wt.w0 = rec ? res.q[0].y : res.q[0].x; wt.w1 = rec ? res.q[0].z : res.q[0].y; wt.w2 = rec ? res.q[0].w : res.q[0].z;
				wt.w3 = rec ? 0u : res.q[0].w; wt.w4 = rec ? 0u : res.q[1].x; wt.w5 = rec ? 0u : res.q[1].y;
				wt.w6 = rec ? 0u : res.q[1].z; wt.w7 = rec ? 0u : res.q[1].w;
				/* the depths that can be compared: up to the end of the query, of the window, of the text (depth == anchor is
				 * the '$' row: nothing to the left) */
				uint32_t lim = L.qlen < wd0 + ncov ? (uint32_t)L.qlen : wd0 + ncov;
				const bool textEnds = anchor < lim;
				if (textEnds) lim = anchor;
				uint32_t tf;
				BT_PROF_T0(t_mm);
				const uint32_t fm = bt_loc_first_mm(L, S, wt, sh, wd0, d, lim, &tf);
				BT_PROF_ADD(PS_LOC_MM, t_mm);
				/* the event: the first mismatch; else a half-and-half boundary; else the text's or the read's end */
				uint32_t dev = fm;
				if (L.halfAndHalf) {
#define BT_LOC_HH(v) do { const uint32_t h_ = (v); if (h_ >= d && h_ < lim && h_ < dev) dev = h_; } while (0)
					BT_LOC_HH(L.d5 - 1u); BT_LOC_HH(L.d5); BT_LOC_HH(L.d3 - 1u); BT_LOC_HH(L.d3);
#undef BT_LOC_HH
				}
				if (dev == 0xffffffffu) {
					if (textEnds) dev = anchor;
					else if (lim == L.qlen) dev = L.qlen - 1u;
				}
				/* what the reference goes through between here and there: one mapLFEx (both rows in one side pair) per
				 * position that is an alternative, one mapLF1 per position that is not */
				const uint32_t upto = dev == 0xffffffffu ? lim : dev;
				{
					BT_PROF_T0(t_tally);
					const uint32_t lo = d > L.fu ? d : (uint32_t)L.fu;
					const uint32_t nalt = bt_loc_count_alt(L, S, lo < upto ? lo : upto, upto);
					req.tally += (nalt << 16) | ((upto - d) - nalt);
					BT_PROF_ADD(PS_LOC_TALLY, t_tally);
				}
				if (dev == 0xffffffffu) {
					/* everything the window holds matches and the read goes on: the next window */
					L.d = upto;
					const uint32_t y = len - anchor + upto;
					BT_REQ_FETCH(WSEL(rtxt) + (y >> 4), ((y & 15u) + (L.qlen - upto) + 63u) >> 6, nullptr);
					L.state = ST_LOC_TXT;
					BT_COUNT_HOST(CN_FETCH);
					return;
				}
				/* the step at the event, as STEP_BEGIN would set it up */
				uint32_t c, q;
				bt_read_get(L, S, L.qlen - dev - 1u, &c, &q);
				L.c = c; L.q = q; L.d = dev;
				const bool alt = (dev >= L.fu) && (!L.considerQuals || (L.ham + bt_mm_penalty(L.maq, q) <= L.qualThresh));
				bool elig = false, over = false;
				if (alt) {
					if (L.considerQuals) {
						if (q < L.lowAltQual) { elig = true; over = true; }
						else if (q == L.lowAltQual) elig = true;
					} else elig = true;
				}
				L.fl_alt = alt; L.fl_elig = elig; L.fl_over = over;
				locMiss = dev == fm || (textEnds && dev == anchor);
				L.lz = (textEnds && dev == anchor) ? 1u : 0u;
				if (dev == fm) L.lt = tf;
				req.tally += alt ? 0x10000u : 1u;
				L.state = ST_STEP_LOC;
			}
			BT_PROF_ADD(PS_LOCUS, t_locus);
		}
		/* ---- resume: the read window arrived ------------------------------------------------- */
		if (!RL && L.state == ST_WIN_DONE) {
			L.cs0 = res.q[0].x; L.cs1 = res.q[0].y; L.cs2 = res.q[0].z; L.cs3 = res.q[0].w;
			L.cq0 = res.x.x; L.cq1 = res.x.y; L.cq2 = res.x.z; L.cq3 = res.x.w;
			L.cchunk = L.scanCb;
			L.state = ST_STEP_BEGIN;
		}
		/* ---- resume: SA walk (reportChaseOne, ebwt.h:2727-2746) --------------------------- */
		if (L.state == ST_CHASE_LFDONE) {
			L.crow = bt_res_lfa(res, BT_RES_ROWL(res));                 /* mapLF(l) */
			L.cjumps++;
			L.state = ST_CHASE_CHECK;
		}
		/* ---- resume: one query position (:456-739) ---------------------------------------- */
		if (L.state == ST_STEP_LFDONE || L.state == ST_STEP_POST || (BT_LOC(RL) && L.state == ST_STEP_LOC)) {
			const uint32_t c = L.c, q = L.q, d = L.d, cur = L.qlen - d - 1u;
			const uint32_t e = bt_ent(L, d);
			bt_row ta[4], tb[4];
			const bool wasLoc = BT_LOC(RL) && L.state == ST_STEP_LOC;
			if (wasLoc) {
				/* decided by the text: the quartet of a one-row range -- the text's base has the one row that goes with it,
				 * the other three ranges are empty (all four at the text's start); nothing is kept of it but the record */
				const uint32_t t = locMiss ? (uint32_t)L.lt : c;
				BT_UNROLL
				for (uint32_t k = 0; k < 4u; k++) { ta[k] = 0; tb[k] = (k == t && !L.lz) ? 1u : 0u; }
				L.bot = locMiss ? L.top : L.top + 1u;
			} else if (L.state == ST_STEP_LFDONE) {
				if (!BT_WIDE && !RL && L.wpf) {
					L.cs0 = res.q[3].x; L.cs1 = res.q[3].y; L.cs2 = res.q[3].z; L.cs3 = res.q[3].w;
					L.cq0 = res.x.x; L.cq1 = res.x.y; L.cq2 = res.x.z; L.cq3 = res.x.w;
					L.cchunk = L.scanCb; L.wpf = 0;
				}
				bt_res_quartets(res, ta, tb);
				const bt_row ac = bt_selr4(c & 3u, ta[0], ta[1], ta[2], ta[3]);
				const bt_row bc = bt_selr4(c & 3u, tb[0], tb[1], tb[2], tb[3]);
				if (L.lfk == LFK_EX2) {
					BT_STORE_ENTRY(e, res);
					if (c < 4u) { L.top = ac; L.bot = bc; }
				} else if (L.lfk == LFK_C2) {
					L.top = ac; L.bot = bc;
				} else {
					/* mapLF1 (ebwt.h:2494-2512) */
					if (BT_RES_ROWL(res) != c || L.top == WSEL(zOff)) { L.top = BT_OFF_MASK; L.bot = BT_OFF_MASK; }
					else { L.top = ac; L.bot = ac + 1u; }
				}
			} else {
				/* no LF was needed: depth-0 fchr quartet (:531-543) or a non-alternative N */
				BT_UNROLL
				for (int k = 0; k < 4; k++) { ta[k] = HFCHR(k); tb[k] = HFCHR(k + 1); }
			}
			/* alternatives at this position (:578-624): the characters other than the read's own whose
			 * range is not empty.  The eliminated-set is the complement of that mask. */
			uint32_t el = (c < 4u) ? (1u << c) : 0u;
			if (L.fl_alt) {
				uint32_t nz = 0;
				BT_UNROLL
				for (uint32_t i = 0; i < 4u; i++) nz |= (tb[i] != ta[i] && i != c) ? (1u << i) : 0u;
				el = ~nz & 15u;
				const uint32_t na = (uint32_t)__builtin_popcount(nz);
				L.altNum = L.altNum + na;
#if BT_L2_TALLY
				if (!L.fl_elig && nz != 0 && L.considerQuals) {
					/* an alternative of a quality above the eligible one: the second level's tallies (BtLane::low2) */
					if (!L.l2v || q < L.low2) { L.low2 = q; L.num2 = na; L.cand2 = d; L.cel2 = el; L.l2v = 1; }
					else if (q == L.low2) { L.num2 = L.num2 + na; L.cand2 = d; L.cel2 = el; }
				}
#endif
				if (L.fl_elig && nz != 0) {
#if BT_L2_TALLY
					if (L.fl_over && L.eligibleNum > 0 && L.considerQuals) {
						/* a lower quality takes over: what was the eligible level is the second one from here on */
						L.low2 = L.lowAltQual; L.num2 = L.eligibleNum; L.cand2 = L.cand; L.cel2 = L.cel1; L.l2v = 1;
					}
					L.cel1 = el;
#endif
					if (L.fl_over) {
						L.lowAltQual = q; L.eligibleNum = 0;
						L.elcint = (nz & 1u) ? 0u : (nz & 2u) ? 1u : (nz & 4u) ? 2u : 3u;
						L.elignore = 0;
					}
					L.eligibleNum = L.eligibleNum + na;
					/* deepest eligible target so far; its ranges go to the LDS candidate slot so that
					 * choosing it later costs no fetch */
					L.cand = d; L.candValid = 1;
					if (!BT_WIDE && S.noCC != 1u) {
						L.ccValid = 1;
						const uint32_t ts = S.tosStride;
						BT_UNROLL
						for (uint32_t k = 0; k < 4u; k++) { S.tos[k * ts] = (uint32_t)ta[k]; S.tos[(4u + k) * ts] = (uint32_t)tb[k]; }
						S.tos[8u * ts] = el | (q << 8);
					}
				}
			}
			META(e) = (uint16_t)(el | (q << 8));
			bool btDespite = false, reportedPartial = false;
			if (cur == 0 && L.top < L.bot && L.sd < L.reportPartials && L.reportPartials > 0) {
				if (L.altNum > 0) btDespite = true;
				if (L.sd > 0) { bt_report_partial(L, S, L.sd); reportedPartial = true; }
			}
			bool invalidExact = false;
			if (cur == 0 && L.sd == 0 && L.bot > L.top && !L.reportExacts) { invalidExact = true; btDespite = true; }
			bool mustBacktrack = false, invalidHH = false, frameFail = false;
			if (L.halfAndHalf) {
				if (d + 1u == L.d5 && L.top < L.bot) {
					invalidHH = (L.sd == 0);
					if (L.sd == 0 && L.altNum > 0) { btDespite = true; mustBacktrack = true; }
					else if (L.sd == 0) frameFail = true;
				} else if (d + 1u == L.d3 && L.top < L.bot) {
					uint32_t lo = 0, hi = 0;
					BT_NOUNROLL
					for (uint32_t i = 0; i < L.sd; i++) {
						uint32_t dd = L.qlen - (BT_FR_MM(i) & 0xffffu) - 1u;
						if (dd < L.d5) hi++; else if (dd < L.d3) lo++;
					}
					invalidHH = (lo == 0 || hi == 0);
					if ((L.sd < 2 || invalidHH) && L.altNum > 0) { mustBacktrack = true; btDespite = true; }
					else if (L.sd < 2) frameFail = true;
				}
			}
			if (frameFail) { L.ret = 0; L.state = ST_FRAME_RETURN; }
			else if (cur == 0 && L.bot > L.top && !invalidHH && !invalidExact && !reportedPartial)
				BT_GOTO_RA(L.sd, L.top, L.bot, L.ham, RC_STEP, L.lmode);
			else if ((L.top == L.bot || btDespite) && L.altNum > 0) {
				if (!(BT_LOC(RL) && wasLoc && locMiss && L.fl_elig && !L.lz && L.top == L.bot && bt_loc_descend<RL>(L, P, W, S, CNT))) L.state = ST_BT_LOOP;
			}
			else if (mustBacktrack || invalidHH || invalidExact || L.top == L.bot) { L.ret = 0; L.state = ST_FRAME_RETURN; }
			else { L.d = d + 1u; L.state = ST_STEP_BEGIN; }
		}
		BT_PROF_ADD(PS_RESUME, t_resume);

		/* ---- everything else ---------------------------------------------------------------- */
		{
			BT_PROF_T0(t_slow);
			if (BT_IS_SLOW(L.state)) bt_lane_slow<RL>(L, P, H, W, C, S, res, req, CNT);
			BT_PROF_ADD(PS_SLOW, t_slow);
		}
		if (req.kind != RQ_NONE) { BT_COUNT_HOST(CN_FETCH); return; }

		/* ---- emit: next query position (:456-568) -------------------------------------------- */
		if (L.state == ST_STEP_BEGIN) {
			if (BT_LOC(RL) && L.lmode) {
				/* a frame in locus mode that goes on (a child just entered, a boundary passed): its next event is found when
				 * the text for it arrives -- one window per call.  Going round this loop again with the window at hand would
				 * save the lane a round and cost its wavefront a trip through everything (scripts/pass_model.py: the
				 * wavefront's trips per round are its slowest lane's) */
				const uint32_t d = L.d;
				if (d >= L.qlen) { L.state = ST_FELL_OFF; return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */ }
				const uint32_t y = (uint32_t)(WSEL(len) - L.top) + d;
				BT_REQ_FETCH(WSEL(rtxt) + (y >> 4), ((y & 15u) + (L.qlen - d) + 63u) >> 6, nullptr);
				L.state = ST_LOC_TXT;
				BT_COUNT_HOST(CN_FETCH);
				return;
			}
			const uint32_t d = L.d;
			if (d >= L.qlen) { L.state = ST_FELL_OFF; return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */ }
			if (L.halfAndHalf && !bt_hh_check_top(L, S, d)) { L.ret = 0; L.state = ST_FRAME_RETURN; return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */ }
			if (L.ebase + (d - L.depth) >= S.a->entCap) { L.state = ST_ABORT; return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */ }
			if (BT_LOC(RL) && W.locOn && !L.hasN && L.top + 1u == L.bot) {
				/* the range is one row: leave row space (its locus record: where the row's suffix is in the text, and the
				 * 48 characters to the left of it).  Reads with an N stay in row space: an N never matches, which the packed
				 * comparison does not know */
				BT_REQ_FETCH(WSEL(loc) + L.top, 1, nullptr);
				L.state = ST_LOC_REC;
				BT_COUNT_HOST(CN_FETCH);
				return;
			}
			uint32_t c, q;
			if (RL) bt_read_get(L, S, L.qlen - d - 1u, &c, &q);
			else if (!bt_window_get(L, L.qlen - d - 1u, &c, &q)) {
				const uint32_t i = L.qlen - d - 1u, j = L.rev ? (L.plen - 1u - i) : i;
				L.scanCb = j >> 4;       /* chunk id, parked until the window arrives */
				BT_REQ_FETCH(H.seq + L.roff + (uint64_t)(j >> 4) * 16u, 1, H.qual + L.roff + (uint64_t)(j >> 4) * 16u);
				L.state = ST_WIN_DONE;
				BT_COUNT_HOST(CN_FETCH);
				return;
			}
			L.c = c; L.q = q;
			const bool alt = (d >= L.fu) && (!L.considerQuals || (L.ham + bt_mm_penalty(L.maq, q) <= L.qualThresh));
			bool elig = false, over = false;
			if (alt) {
				if (L.considerQuals) {
					if (q < L.lowAltQual) { elig = true; over = true; }
					else if (q == L.lowAltQual) elig = true;
				} else elig = true;
			}
			L.fl_alt = alt; L.fl_elig = elig; L.fl_over = over;
			const bt_row rtop = L.top, rbot = L.bot;
			if (c == 4u && d > 0) { L.top = 1; L.bot = 1; }
			if (rtop == 0 && rbot == 0) {
				/* depth 0: the fchr quartet (:531-543) */
				const uint32_t e = L.ebase + (d - L.depth);
				const bt_row f0 = HFCHR(0), f1 = HFCHR(1), f2 = HFCHR(2), f3 = HFCHR(3), f4 = HFCHR(4);
				bt_store_quartet(PT4(e), f0, f1, f2, f3);
				bt_store_quartet(PB4(e), f1, f2, f3, f4);
				if (c < 4u) { L.top = bt_selr4(c, f0, f1, f2, f3); L.bot = bt_selr4(c, f1, f2, f3, f4); }
				L.state = ST_STEP_POST;
				return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */
			} else if (alt || c < 4u) {
				if (alt) { BT_REQ_RANK2(rtop, rbot); L.lfk = LFK_EX2; }
				else if (L.top + 1u == L.bot) { BT_REQ_RANK1(L.top); L.lfk = LFK_LF1; }
				else { BT_REQ_RANK2(L.top, L.bot); L.lfk = LFK_C2; }
				/* the next position (d+1) leaves the 16-base window: fetch the neighbouring chunk with
				 * this round's rank request instead of spending a round on it */
				L.wpf = 0;
				if (!BT_WIDE && !RL && d + 1u < L.qlen) {
					const uint32_t i2 = L.qlen - d - 2u, j2 = L.rev ? (L.plen - 1u - i2) : i2;
					if ((j2 >> 4) != L.cchunk) { req.wchunk = j2 >> 4; L.scanCb = j2 >> 4; L.wpf = 1; }
				}
				L.state = ST_STEP_LFDONE; return;
			} else {
				/* non-alternative N: the range is already (1,1); only the bookkeeping remains */
				L.state = ST_STEP_POST;
				return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */
			}
		}
		/* ---- emit: next SA-walk step, or the SA sample once the walk has arrived ---------------- */
		if (L.state == ST_CHASE_CHECK) {
			if (!BT_WIDE && W.locOn) {
				/* the dense suffix array: the row's locus record instead of the walk to a sampled row */
				BT_REQ_FETCH(WSEL(loc) + L.crow, 1, nullptr);
				L.state = ST_RESOLVE_DONE;
				BT_COUNT_HOST(CN_FETCH);
				return;
			}
			if ((L.crow & WSEL(offMask)) != L.crow && L.crow != WSEL(zOff)) {
				BT_REQ_RANK1(L.crow); L.lfk = LFK_CHASE;
				L.state = ST_CHASE_LFDONE; return;
			}
			L.state = ST_RESOLVE_DONE;
			if (L.crow != WSEL(zOff)) {
				const bt_row* offs = WSEL(offs);
				BT_REQ_FETCH(offs + ((L.crow >> WSEL(offRate)) & ~(bt_row)(BT_PIECE_ROWS - 1u)), 1, nullptr);
				BT_COUNT_HOST(CN_FETCH);
				return;
			}
			return;        /* (the lane goes on in the next round's pass: see the note at the head of the loop) */
		}
		if (L.state == ST_IDLE) return;
	}
}

#undef FRW
#undef PT
#undef PB
#undef META
#undef PALS
#undef IXSEL
#undef HSEL
#undef WSEL
#undef HFCHR
#undef ST_IS
#undef ST_IS_NOREQ
#endif /* BT_CORE_H_ */
