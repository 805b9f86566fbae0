/* bt_kernels.h -- kernel argument block and launchers shared by bt_kernels.hip and bt_api.cpp */
#ifndef BT_KERNELS_H_
#define BT_KERNELS_H_

#include "bt_core.h"
#include "bt_best.h"

#define BT_BLOCK 256

/* Carry-over.  When the read cursor of a launch runs dry, every lane is in the middle of a read, and the reads
 * that backtrack for 10^5 rounds would keep their wavefronts -- and through them the workgroups' LDS -- for
 * seconds after the rest have finished (the "tail": at 16 M reads per launch more than half the launch).
 * Instead the lanes park what they are doing: lane g writes its whole state (automaton + pending request) to
 * pool[g], the wavefront exits, and lane g of the context's next launch -- same grid, same scratch slot g --
 * picks it up again before taking fresh reads.  A read stays with its lane until it is done, through as many
 * launches as it takes up to `maxAge`; a read that old is no longer parked but finished by the launch it is in.
 * So the results of batch k are complete when launch k + maxAge is, or after the flush launch bt_ctx_sync
 * enqueues; which batches still have reads parked after a launch is in parkedOf[]. */
#if BT_WIDE
#define BT_POOL_WORDS 80     /* the wide build's lane state is up to 17 pieces */
#define BT_POOL_REQ 17
#else
#define BT_POOL_WORDS 64
#define BT_POOL_REQ 13
#endif
/* in 16-byte pieces: [0, BT_POOL_REQ) BtLane, [BT_POOL_REQ] request kind / n / wchunk, [BT_POOL_REQ + 1] request a, x,
 * [BT_POOL_REQ + 2] word 0: the stamp (narrow build: words [0..51], [52..54], [56..59], [60]) */
struct BtPoolRec { uint32_t w[BT_POOL_WORDS]; };
#define BT_POOL_STAMP_WORD (4 * (BT_POOL_REQ + 2))

struct BtKernelArgs {
	BtHot      H;                /* by value: scalar registers                                   */
	const BtCold* cold;          /* device memory: program, full index descriptors, batch        */
	const BtWarm* warm;          /* device memory; each workgroup copies it to LDS               */
	/* per-slot scratch arenas (see BtScratch): lane g works in slot g                              */
	uint32_t*  frames;           /* [nSlots][frCap][12]                                          */
	uint32_t*  pairs;            /* [nSlots][entCap][8]                                          */
	uint16_t*  meta;             /* [nSlots][entCap] mask | Phred<<8                             */
	uint64_t*  pals;             /* [nSlots][palCap]                                             */
	uint32_t   nLanes, nSlots, frCap, entCap, palCap;
	uint32_t*  nextRead;         /* work cursor: read ids                                        */
	const uint32_t* order;       /* optional: read id for each cursor value                      */
	/* carry-over (see above) */
	BtPoolRec* pool;             /* [nLanes]; NULL = none                                        */
	uint32_t   launchSeq;        /* this launch's number on its context; a record is live if stamped launchSeq - 1 */
	uint32_t   adopt, park;      /* pick parked reads up at the start / park at the end          */
	uint32_t   maxAge;           /* launches a read may be carried through (< BT_BATCH_RING - 1) */
	uint32_t   parkMinRounds;    /* diagnostics (BT_PARK_MIN_ROUNDS): a wavefront parks no earlier than its round N */
	uint32_t*  parkedOf;         /* [BT_BATCH_RING] reads parked by this launch, per batch-ring slot */
	const uint32_t* orderCount;  /* non-null (with order): the pick-up list's length lives on the device -- min(*orderCount,
	                                orderCap) entries.  The on-stream second pass over reads that outgrew their scratch */
	uint32_t   orderCap;
	const uint32_t* gate;        /* non-null: run only if gateLo <= *gate <= gateHi.  Lets the host enqueue both
	                                builds of the kernel for a batch whose longest read is only known on the
	                                device (bt_align_batch_device) without waiting for it                 */
	uint32_t   gateLo, gateHi;
	unsigned long long* counts;  /* CN_N x u64 = bt_op_counts                                    */
#ifdef BT_TRACE
	/* diagnostics build only (make -C bowtie_amd/csrc trace): every round of the lane that holds read `traceRead`
	 * appends 12 words to trace[] (trace[0] = records written): round, state, step|mirror|readFw|rev, request kind,
	 * n, a lo/hi, x lo/hi, top, bot, d|sd */
	uint32_t*  trace; uint32_t traceRead, traceCap;
#endif
};

/* the best-first kernel (bt_best_kernels.hip) */
struct BtBestArgs {
	const BfProgram*  prog;      /* device memory */
	const BtIndexDev* ix;        /* [2] */
	const BtBatchDev* batch;
	const BtRefDev*   ref;       /* paired-end: the 2-bit reference (device memory); NULL otherwise */
	uint32_t*  arenas;           /* [nLanes][arenaWords] */
	uint32_t   arenaWords;
	uint32_t*  nextRead;
	unsigned long long* counts;
	/* second pass over the reads that outgrew their arena in the first: their ids and how many */
	const uint32_t* workList; const uint32_t* workCount; uint32_t workCap;
	/* which loop runs the reads (bt_best_kernels.hip): 1 = every lane its read from start to finish (PairedBWAlignerV1
	 * always; BT_BEST_NESTED=1 for the rest), 0 = the wavefront automaton of bt_best.h, with its gates: a cold sweep when
	 * coldMin lanes wait for one, new reads when takeMin lanes wait for one, ended streaks finished every sendPeriod-th
	 * round or when sendMin lanes wait for it */
	uint32_t laneStride;   /* > 1: only every laneStride-th lane of a block takes reads, arena g / laneStride (the second pass: its few
	                        * hundred heavy reads over many wavefronts instead of 64 to a wavefront) */
	uint32_t nested, coldMin, takeMin, sendPeriod, sendMin, sweepTwice;   /* sweepTwice: a second pass of a sweep for the lanes its first left in the middle */
};

extern "C" {
/* ids of the reads whose status has `flag` set -> list[0 .. *count) (order unspecified) */
int bt_launch_collect_flagged(const uint8_t* status, uint32_t n, uint32_t flag, uint32_t* list, uint32_t* count, uint32_t cap, void* stream);
int bt_launch_best(const BtBestArgs* a, uint32_t nBlocks, void* stream);
uint32_t bt_best_blocks_per_cu(int nested);      /* blocks per CU the call-by-call kernel (1) / the wavefront automaton (0) was built for */
int bt_launch_jump_build(const BtIndexDev* ix, uint32_t K, uint32_t* jump, uint16_t* meta, void* stream);
#define BT_RL_FORCE_EXT 0x100     /* or'ed into `rl`: launch the EXT instance whatever the arguments ask for (diagnostics) */
int bt_launch_search(const BtKernelArgs* a, uint32_t nBlocks, int occ, int rl, void* stream);
int bt_launch_maxlen(const uint16_t* len, uint32_t n, uint32_t* out, void* stream);   /* *out = max(*out, max len[]) */
int bt_launch_gather_bench(const BtIndexDev* ix, uint32_t nBlocks, uint32_t iters, uint32_t dep, uint32_t* sink, void* stream);
int bt_launch_probe_rank(const BtIndexDev* ix, const bt_row* rows, uint32_t n, bt_row* lf,
                         uint8_t* L, uint32_t sides, void* stream);
int bt_launch_blk_build(const BtIndexDev* ix, uint8_t* out, uint32_t nBlocks, void* stream);
int bt_launch_loc_build(const BtIndexDev* ix, BtU4* loc, uint32_t* rtxtAlloc, uint16_t* walk, void* stream);
int bt_launch_probe_chase(const BtIndexDev* ix, const bt_row* rows, uint32_t n, uint32_t qlen,
                          bt_row* joined, uint32_t* tidx, uint32_t* toff, void* stream);
}
#endif
