#!/bin/bash
# Final GPU call of round 5, most important first: the default command exactly as the driver runs it (CPU baseline, every
# other workload of --also auto); the whole GPU suite on the final code; rocprofv3 --kernel-trace --stats of the default
# workload; PMC FETCH / WRITE of the final search kernel at 64 M reads per launch (profiles/traffic.json).
#   gpurun --timeout 1500 -- 'bash scripts/r5/final.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_final; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log
say "python bench.py (the default command), $(( $(date +%s) - t0 )) s wall:"
python - "$O/bench_default.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c, cb = d["roofline"], d["config"], d.get("cpu_baseline", {})
print("   value %.3f M aligned reads/s, %.3f M reads processed/s, %.1f ms/step; roofline: %s %.1f / %.0f %s = frac %.4f, kernel %s avg %.1f ms, traffic %s; gather ceiling %.0f GB/s (%s)" % (
    d["value"] / 1e6, d["reads_processed_per_s"] / 1e6, d["ms_per_step"], r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"], r["kernel"], r["kernel_ms_avg"], r.get("traffic"),
    r.get("gather_ceiling_GBps", 0), r.get("gather_ceiling_source")))
print("   verified %s %s; diffed vs reference %s reads, %s mismatches; cpu_baseline %.1f k reads/s (%s, %s cores)" % (c.get("hits_verified_against_text"), c.get("verified_unit"), c.get("reads_diffed_vs_reference"), c.get("diff_mismatches"), cb.get("value", 0) / 1e3, cb.get("kind"), cb.get("cores")))
for k, v in (c.get("other_workloads") or {}).items():
    if "error" in v: print("   other workload %-20s ERROR %s" % (k, v["error"][-200:])); continue
    print("   other workload %-20s %.3f M reads processed/s (%.3f M aligned), frac %.4f, %s; diffed %s, mismatches %s" % (k, (v.get("reads_processed_per_s") or 0) / 1e6, v["value"] / 1e6, v["roofline_frac"], v["kernel"], v.get("reads_diffed_vs_reference"), v.get("diff_mismatches")))
PY
timeout 480 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
say "pytest -m gpu (whole suite, final code, six workers): $(tail -1 $O/gpu_suite.txt)"
grep -h "^FAILED" $O/gpu_suite.txt | head -5 | tee -a $S
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --also none > $O/bench_trace.json 2> $O/bench_trace.log
python - "$O/trace_default" "$O/bench_trace.json" >> $S <<'PY'
import sys, csv, glob, json
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 5: print("   rocprofv3 kernel stats:", row)
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("   bench.py under rocprofv3: %.3f M reads/s, kernel %s avg %.1f ms (HIP events), frac %.4f" % (d["reads_processed_per_s"] / 1e6, d["roofline"]["kernel"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c --kernel-include-regex bt_search --output-format csv -d $O/pmc_$c -- python $R/bench.py --reads 64000000 --no-carry --steps 1 --warmup 0 --no-cpu --no-verify --also none > $O/pmc_$c.json 2> $O/pmc_$c.log
	python - "$O/pmc_$c" "$c" >> $S <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bt_search" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("   PMC %s of bt_search_kernel, 64 M reads per launch, per dispatch: %s" % (k, ["%.4g" % x for x in v]))
PY
done
cd $R
cat $S
