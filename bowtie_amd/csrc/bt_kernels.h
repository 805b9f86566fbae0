/* bt_kernels.h -- kernel argument block and launchers shared by bt_kernels.hip and bt_api.cpp */
#ifndef BT_KERNELS_H_
#define BT_KERNELS_H_

#include "bt_core.h"
#include "bt_best.h"

#define BT_BLOCK 256

/* Carry-over.  When the read cursor of a launch runs dry, every lane is in the middle of a read, and the reads
 * that backtrack for 10^5 rounds would keep their wavefronts -- and through them the workgroups' LDS -- for
 * seconds after the rest have finished (the "tail": at 16 M reads per launch more than half the launch).
 * Instead the lanes park what they are doing: the lane's whole state (automaton, pending request, scratch slot)
 * goes into a pool record, the wavefront exits, and the next launch on the same context picks the parked reads
 * up first, mixed in with its own fresh reads.  A read carried into a launch is finished by that launch (it is
 * not parked a second time), so the results of batch k are complete when launch k+1 is, or after the flush
 * launch bt_ctx_sync enqueues. */
#define BT_POOL_WORDS 64
struct BtPoolRec { uint32_t w[BT_POOL_WORDS]; };   /* [0..47] BtLane, [48] slot, [52..53] request kind/n, [56..59] request a/x */

struct BtKernelArgs {
	BtHot      H;                /* by value: scalar registers                                   */
	const BtCold* cold;          /* device memory: program, full index descriptors, batch        */
	const BtWarm* warm;          /* device memory; each workgroup copies it to LDS               */
	/* per-slot scratch arenas (see BtScratch).  Lane g of a launch works in slot slotBase + g; a read carried over
	 * from the previous launch keeps the slot it was parked with (the launches of a context alternate between
	 * two sets of slots)                                                                           */
	uint32_t*  frames;           /* [nSlots][frCap][12]                                          */
	uint32_t*  pairs;            /* [nSlots][entCap][8]                                          */
	uint16_t*  meta;             /* [nSlots][entCap] mask | Phred<<8                             */
	uint64_t*  pals;             /* [nSlots][palCap]                                             */
	uint32_t   nLanes, nSlots, frCap, entCap, palCap, slotBase;
	uint32_t*  nextRead;         /* work cursor: read ids                                        */
	const uint32_t* order;       /* optional: read id for each cursor value                      */
	/* carry-over (see above): reads parked by the previous launch, and where this one parks its own */
	const BtPoolRec* carryIn; const uint32_t* carryInCount; uint32_t* carryCursor;
	BtPoolRec* carryOut;      uint32_t* carryOutCount;      uint32_t carryOutCap;
	const uint8_t* prevSeq; const uint8_t* prevQual; uint32_t prevStride;   /* the previous batch's reads (cold->B[1] has the rest) */
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
int bt_launch_gather_bench(const BtIndexDev* ix, uint32_t nBlocks, uint32_t iters, uint32_t dep, uint32_t* sink, void* stream);
int bt_launch_probe_rank(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t* lf,
                         uint8_t* L, void* stream);
int bt_launch_probe_chase(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t qlen,
                          uint32_t* joined, uint32_t* tidx, uint32_t* toff, void* stream);
}
#endif
