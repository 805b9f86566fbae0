#!/bin/bash
# Round 3, fourth GPU call: the 32-byte rank blocks (bt_rank.h) -- probe tests (blocks == side layout == reference's
# known answers), parity subset, A/B against the previous kernel (libbowtie_amd_base.so) on the hg19-scale index.
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "probe or ragged or carry or stream or idempot or 1024 or overflow or retries or golden_sam" > $O/parity.txt 2>&1; say "parity subset: $(tail -1 $O/parity.txt)"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, kernel %.1f ms, rounds/read %.1f, frac %.4f' % (d['value']/1e6, r.get('kernel_ms_avg', 0), r.get('lane_iters_per_read',0), r['frac']))" 2>&1 | tail -1; }
for lib in libbowtie_amd_base.so libbowtie_amd.so; do
  f=$O/bench_16M_${lib%.so}
  BT_LIB=$lib timeout 400 python bench.py --workload big_n2_100 --reads 16000000 --steps 6 --warmup 2 --no-cpu --no-verify > $f.json 2> $f.log
  say "big_n2_100 16M carry-over $lib: $(val $f.json)"
done
f=$O/bench_64M_libbowtie_amd
timeout 400 python bench.py --workload big_n2_100 --reads 64000000 --steps 2 --warmup 1 --no-cpu > $f.json 2> $f.log
say "big_n2_100 64M no carry libbowtie_amd.so (base: 8.02 M reads/s in r3d): $(val $f.json)"
python - >> $S 2>&1 <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bowtie_amd import aligner as AL, _abi as A
import ctypes as C, os
from bowtie_amd import ebwt_build as EB
import torch
base, text, note = EB.ensure_big_index(0, torch.device("cuda", 0))
idx = AL.Index(base, need_mirror=True)
al = AL.Aligner(idx, A.make_policy(mode="v", mms=0))
lib = AL.lib()
lib.bt_bench_gather.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
for sides in (0, 2):
    for dep in (0, 1):
        for nb in (1024, 4096):
            ms, gbs = C.c_float(), C.c_double()
            lib.bt_bench_gather(al._h, sides, nb, 256, dep, C.byref(ms), C.byref(gbs))
            q = nb * 256 * 256 / (ms.value * 1e-3) / 1e9
            print("gather ceiling %s dep=%d blocks=%d: %.2f ms, %.1f G queries/s, %.2f TB/s of %d-byte units" % ("side pairs" if sides else "rank blocks", dep, nb, ms.value, q, gbs.value / 1e3, 128 if sides else 32))
PY
cat $S
