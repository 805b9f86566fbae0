/* bt_kernels.h -- kernel argument block and launchers shared by bt_kernels.hip and bt_api.cpp */
#ifndef BT_KERNELS_H_
#define BT_KERNELS_H_

#include "bt_core.h"

#define BT_BLOCK 256

struct BtKernelArgs {
	BtHot      H;                /* by value: scalar registers                                   */
	const BtCold* cold;          /* device memory: program, full index descriptors, batch        */
	/* per-lane scratch arenas (see BtScratch) */
	uint32_t*  frames;           /* [nLanes][frCap][16]                                           */
	uint32_t*  pairs;            /* [nLanes][entCap][8]                                          */
	uint16_t*  meta;             /* [nLanes][entCap] mask | Phred<<8                             */
	uint64_t*  pals;             /* [nLanes][palCap]                                             */
	uint32_t   nLanes, frCap, entCap, palCap;
	uint32_t*  nextRead;         /* global read cursor                                           */
	unsigned long long* counts;  /* CN_N x u64 = bt_op_counts                                    */
};

extern "C" {
int bt_launch_search(const BtKernelArgs* a, uint32_t nBlocks, int occ, void* stream);
int bt_launch_probe_rank(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t* lf,
                         uint8_t* L, void* stream);
int bt_launch_probe_chase(const BtIndexDev* ix, const uint32_t* rows, uint32_t n, uint32_t qlen,
                          uint32_t* joined, uint32_t* tidx, uint32_t* toff, void* stream);
}
#endif
