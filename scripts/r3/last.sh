#!/bin/bash
# Round 3, a short last call: the headline configuration as the driver will run it (no CPU leg), config 3 without
# carry-over, and FETCH_SIZE calibrated on known-count gathers of both index layouts.
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (aligned %.3f M), kernel %.1f ms, frac %.4f, %s' % (d['value']/1e6, d['aligned_reads_per_s']/1e6, r.get('kernel_ms_avg', 0), r['frac'], r['kernel']))" 2>&1 | tail -1; }
f=$O/bench_default_nocpu; timeout 400 python bench.py --no-cpu --also none > $f.json 2> $f.log; say "default (big_n2_100, 200 M reads per step, 2 steps): $(val $f.json)"
f=$O/bench_big_v2_76; timeout 300 python bench.py --workload big_v2_76 --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_v2_76 (50 M reads per step, no carry-over): $(val $f.json)"
cd /tmp
for lay in blocks sides; do
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bt_gather" --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_gather_$lay -- python $GRAFT_REPO_ROOT/scripts/gather_pmc_probe.py $lay > $GRAFT_REPO_ROOT/$O/gather_$lay.log 2>&1
  python - $GRAFT_REPO_ROOT/$O/pmc_gather_$lay $lay >> $GRAFT_REPO_ROOT/$S <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    q = 4096 * 256 * 256
    print("FETCH_SIZE on %s gathers: %s KB per launch of %d queries = %s bytes per query as tallied" % (sys.argv[2], ["%.4g" % x for x in v], q, ["%.1f" % (x * 1024 / q) for x in v]))
PY
done
cd $GRAFT_REPO_ROOT; grep -h "queries per launch" $O/gather_*.log | tee -a $S
cat $S
