/* bt_kernels.h -- kernel argument block and launchers shared by bt_kernels.hip and bt_api.cpp */
#ifndef BT_KERNELS_H_
#define BT_KERNELS_H_

#include "bt_core.h"
#include "bt_best.h"

#define BT_BLOCK 256

/* A read that has run for more than `heavyRounds` rounds is taken out of its lane: the lane's whole
 * state (automaton + pending request + scratch slot) is parked in a pool record and the lane pulls
 * the next read.  A follow-up launch of the same kernel adopts the parked reads, one per lane, so
 * the few reads that backtrack for 10^5 rounds neither hold 63 idle lanes hostage nor keep the
 * batch's other wavefronts from retiring. */
#define BT_POOL_WORDS 64
struct BtPoolRec { uint32_t w[BT_POOL_WORDS]; };   /* [0..47] BtLane, [48] slot, [52..53] request kind/n, [56..59] request a/x */

struct BtKernelArgs {
	BtHot      H;                /* by value: scalar registers                                   */
	const BtCold* cold;          /* device memory: program, full index descriptors, batch        */
	const BtWarm* warm;          /* device memory; each workgroup copies it to LDS               */
	/* per-slot scratch arenas (see BtScratch); slots 0..nLanes-1 belong to the lanes of the first
	 * launch, the rest are handed out when a lane parks a heavy read and needs a fresh slot      */
	uint32_t*  frames;           /* [nSlots][frCap][12]                                          */
	uint32_t*  pairs;            /* [nSlots][entCap][8]                                          */
	uint16_t*  meta;             /* [nSlots][entCap] mask | Phred<<8                             */
	uint64_t*  pals;             /* [nSlots][palCap]                                             */
	uint32_t   nLanes, nSlots, frCap, entCap, palCap;
	uint32_t*  nextRead;         /* work cursor: read ids (level 0) or pool records (level > 0)  */
	const uint32_t* order;       /* optional: read id for each cursor value (heavy-first schedule) */
	uint32_t*  nextSlot;         /* spare-slot cursor (starts at nLanes)                         */
	const BtPoolRec* poolIn;  const uint32_t* poolInCount;      /* NULL at level 0               */
	BtPoolRec* poolOut;       uint32_t* poolOutCount;  uint32_t poolOutCap;   /* NULL at the last level */
	uint32_t   heavyRounds;      /* park reads that reach this many rounds                       */
	const uint32_t* orderCount;  /* non-null (with order): the pick-up list's length lives on the device -- min(*orderCount,
	                                orderCap) entries.  The on-stream second pass over reads that outgrew their scratch */
	uint32_t   orderCap;
	const uint32_t* gate;        /* non-null: run only if gateLo <= *gate <= gateHi.  Lets the host enqueue both
	                                builds of the kernel for a batch whose longest read is only known on the
	                                device (bt_align_batch_device) without waiting for it                 */
	uint32_t   gateLo, gateHi;
	unsigned long long* counts;  /* CN_N x u64 = bt_op_counts                                    */
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
};

extern "C" {
/* ids of the reads whose status has `flag` set -> list[0 .. *count) (order unspecified) */
int bt_launch_collect_flagged(const uint8_t* status, uint32_t n, uint32_t flag, uint32_t* list, uint32_t* count, uint32_t cap, void* stream);
int bt_launch_best(const BtBestArgs* a, uint32_t nBlocks, void* stream);
int bt_launch_search(const BtKernelArgs* a, uint32_t nBlocks, int occ, int rl, void* stream);
int bt_launch_maxlen(const uint16_t* len, uint32_t n, uint32_t* out, void* stream);   /* *out = max(*out, max len[]) */
int bt_launch_schedule(const uint8_t* seq, const uint16_t* len, uint32_t stride, uint32_t n,
                       const uint32_t* ftab, uint32_t ftabChars, uint32_t textLen,
                       uint8_t* bucket, uint32_t* hist, uint32_t* order, void* stream);
int bt_launch_gather_bench(const BtIndexDev* ix, uint32_t nBlocks, uint32_t iters, uint32_t dep, uint32_t* sink, void* stream);
int bt_launch_probe_rank(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t* lf,
                         uint8_t* L, void* stream);
int bt_launch_probe_chase(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t qlen,
                          uint32_t* joined, uint32_t* tidx, uint32_t* toff, void* stream);
}
#endif
