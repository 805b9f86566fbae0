#!/bin/bash
# Round 3, second GPU call: what round 2 left unmeasured about bt_search_kernel on the hg19-scale index.
#   gpurun --timeout 1500 -- 'bash scripts/r3/traffic.sh'
# 1. PMC passes on a 16 M-read launch (scripts/prof.sh: FETCH_SIZE / WRITE_SIZE in their own passes, then the SQ groups),
#    never combined with sys/hip/hsa traces -> HBM bytes per read (roofline.traffic; apply the gfx950 FETCH_SIZE correction
#    of profiles/r1_final/calib_fetch_size.txt) and VALU/SALU instructions per wave-round.
# 2. The same launch size with carry-over (bench.py --carry 12 --pipes 1) against without: the batch-size cliff on the
#    rebuilt EXT instances.
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
SKIP_TRACE=1 bash scripts/prof.sh r3b_pmc --workload big_n2_100 --steps 1 --warmup 1 --reads 16000000 --pipes 1 > $O/pmc_summary.txt 2>&1
tail -40 $O/pmc_summary.txt
for c in 0 12; do
  timeout 600 python bench.py --workload big_n2_100 --reads 16000000 --steps 6 --warmup 2 --carry $c --no-cpu --no-verify > $O/bench_16M_carry$c.json 2> $O/bench_16M_carry$c.log
  python -c "import json; d=json.loads(open('$O/bench_16M_carry$c.json').read().strip().splitlines()[-1]); print('carry $c: %.2f M reads/s' % (d['value']/1e6))"
done
