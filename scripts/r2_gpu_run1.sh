#!/bin/bash
# round-2 GPU pass 1: full GPU test suite, the default bench line, and the drain-parking A/B at 16 M-read launches
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -5 $O/gputests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
for P in 0 8 16 32; do
  BT_DRAIN_PARK=$P timeout 300 python bench.py --reads 16000000 --pipes 2 --steps 6 --warmup 2 --no-cpu --no-verify > $O/bench_16m_park$P.json 2> $O/bench_16m_park$P.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/bench_16m_park$P.json").read().strip().splitlines()[-1]); print("park$P 16M x2 pipes:", j["value"], j["ms_per_step"])
except Exception as e: print("park$P failed", e)
PY
done
for P in 0 16; do
  BT_DRAIN_PARK=$P timeout 300 python bench.py --reads 16000000 --pipes 1 --steps 6 --warmup 2 --no-cpu --no-verify > $O/bench_16m_p1_park$P.json 2>&1
  tail -c 400 $O/bench_16m_p1_park$P.json | head -c 200; echo
done
BT_DRAIN_PARK=16 timeout 400 python bench.py --no-cpu --no-verify > $O/bench_200m_park16.json 2> $O/bench_200m_park16.err; head -c 300 $O/bench_200m_park16.json
