#!/usr/bin/env python3
"""Known-count gathers for calibrating rocprofv3's FETCH_SIZE on the rank-block access pattern:
   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex bt_gather -- python scripts/gather_pmc_probe.py [sides]
runs bt_bench_gather (2 launches of n_blocks x 256 lanes x iters queries each) on the hg19-scale index and prints the
query count per launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bowtie_amd import aligner as AL, _abi as A, ebwt_build as EB
base, text, note = EB.ensure_big_index(0, torch.device("cuda", 0))
idx = AL.Index(base, need_mirror=True)
al = AL.Aligner(idx, A.make_policy(mode="v", mms=0))
lib = AL.lib()
lib.bt_bench_gather.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
sides = 2 if (len(sys.argv) > 1 and sys.argv[1] == "sides") else 0
ms, gbs = C.c_float(), C.c_double()
nb, it = 4096, 256
lib.bt_bench_gather(al._h, sides, nb, it, 0, C.byref(ms), C.byref(gbs))
print("layout %s: %d queries per launch, %.2f ms" % ("sides" if sides else "blocks", nb * 256 * it, ms.value))
