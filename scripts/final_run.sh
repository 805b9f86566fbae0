#!/bin/bash
# Round-end consolidation on the GPU box (from the repo root): GPU tests, smoke, the default bench
# line under rocprofv3 --kernel-trace --stats (same command), PMC passes on a 16 M-read launch,
# and the other BASELINE configs.  Everything lands in gpurun_out/final/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err )
tail -c 600 $O/bench_default.json
for w in ecoli_v0_36 ecoli_v2_76 ecoli_n2_100 big_v2_76; do
  python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err; cut -c1-160 $O/bench_$w.json
done
SKIP_TRACE=1 bash scripts/prof.sh final_pmc --workload big_n2_100 --steps 1 --warmup 1 --reads 16000000 --pipes 1 > $O/pmc_summary.txt 2>&1
tail -30 $O/pmc_summary.txt
find $O/trace_default -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_default.csv \;
cat $O/kernel_stats_default.csv | head -5
