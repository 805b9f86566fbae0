#!/bin/bash
# Last GPU call of round 6: the tree left on main (against the fourth final call: a comment in include/bowtie_amd.h, documents, and
# the binaries rebuilt for it) -- the driver's test command, smoke(), the default workload for two steps.
#   gpurun --timeout 900 -- 'bash scripts/r6/final5.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_final5; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
t0=$(date +%s)
timeout 700 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite_x.txt 2>&1
say "python -m pytest tests/ -x -q -m gpu (the driver's command), $(( $(date +%s) - t0 )) s: $(tail -1 $O/gpu_suite_x.txt)"
grep -h "^FAILED" $O/gpu_suite_x.txt | head -5 | tee -a $S
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
say "smoke(): $(tail -1 $O/smoke.txt)"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $O/bench.json 2> $O/bench.log
python - "$O/bench.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("python bench.py --steps 2 --warmup 1 --no-cpu --also none: %.3f M reads processed/s, kernel %s avg %.1f ms, frac %.4f, traffic %s, verified %s" % (d["reads_processed_per_s"] / 1e6, r["kernel"], r["kernel_ms_avg"], r["frac"], r.get("traffic"), d["config"].get("hits_verified_against_text")))
except Exception as e:
    print("bench: FAILED (%s)" % e)
PY
cat $S
