#!/usr/bin/env python3
"""Random-128-byte-gather ceiling of the GPU on an index: bt_bench_gather over a
range of lanes-in-flight, independent and dependent (SA-walk-like) queries.  Prints JSON.

    python scripts/gather_ceiling.py [--index big|ecoli] [--genome BP]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bowtie_amd import _abi as A  # noqa: E402
from bowtie_amd import aligner as AL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", default="big")
    ap.add_argument("--genome", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.index == "ecoli":
        base = os.path.join(ROOT, "tests", "golden", "e_coli")
    else:
        from bowtie_amd import ebwt_build as EB
        base, _, _ = EB.ensure_big_index(args.genome, dev, 0, 1)
    idx = AL.Index(base, need_mirror=True, device=0)
    al = AL.Aligner(idx, A.make_policy())
    lib = AL.lib()
    lib.bt_bench_gather.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                    C.POINTER(C.c_float), C.POINTER(C.c_double)]
    out = {"index": base, "ebwt_bytes": int(idx.info.ebwt_bytes), "runs": []}
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for dep in (0, 1):
        for blocks_per_cu in (1, 2, 4, 8):
            ms, gbs = C.c_float(), C.c_double()
            nb = cus * blocks_per_cu
            iters = 4096 if dep == 0 else 1024
            rc = lib.bt_bench_gather(al._h, 0, nb, iters, dep, C.byref(ms), C.byref(gbs))
            if rc != 0:
                raise SystemExit("bt_bench_gather rc=%d" % rc)
            out["runs"].append({"dependent": bool(dep), "lanes": nb * 256, "blocks_per_cu": blocks_per_cu,
                                "queries": nb * 256 * iters, "ms": ms.value, "GBps_128B_per_query": gbs.value,
                                "Gqueries_per_s": gbs.value / 128.0})
    out["ceiling_GBps"] = max(r["GBps_128B_per_query"] for r in out["runs"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
