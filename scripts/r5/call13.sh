#!/bin/bash
# Follow-up to the final call: (1) test_gpu_host_batches_streamed_finished_by_ticks[1] failed once in the whole-suite run (six
# workers): what failed, and does it again -- alone, and under load; (2) the default command once more, now that bench.py frees
# its index before it starts the other workloads' processes (in the final call they found the device too full for the locus image).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_13; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
for k in 1 2 3; do
	timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 0 -k "finished_by_ticks" > $O/ticks_alone_$k.txt 2>&1
	say "finished_by_ticks alone, run $k: $(tail -1 $O/ticks_alone_$k.txt)"
done
for k in 1 2; do
	timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "streamed or stress or carry or locus or ragged" > $O/ticks_load_$k.txt 2>&1
	say "streamed / carry-over / locus / ragged tests under six workers, run $k: $(tail -1 $O/ticks_load_$k.txt)"
	grep -h "^FAILED" $O/ticks_load_$k.txt | head -3 | tee -a $S
done
grep -h -B2 -A14 "AssertionError\|assert " $O/ticks_*.txt | head -60 >> $S
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log
say "python bench.py (the default command), $(( $(date +%s) - t0 )) s wall:"
python - "$O/bench_default.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c, cb = d["roofline"], d["config"], d.get("cpu_baseline", {})
print("   value %.3f M aligned reads/s, %.3f M reads processed/s, %.1f ms/step; frac %.4f, kernel %s avg %.1f ms, traffic %s (as tallied %s)" % (
    d["value"] / 1e6, d["reads_processed_per_s"] / 1e6, d["ms_per_step"], r["frac"], r["kernel"], r["kernel_ms_avg"], r.get("traffic"), r.get("traffic_as_tallied")))
print("   verified %s; diffed vs reference %s reads, %s mismatches; cpu_baseline %.1f k reads/s (%s, %s cores; -p 1: %s)" % (c.get("hits_verified_against_text"), c.get("reads_diffed_vs_reference"), c.get("diff_mismatches"), cb.get("value", 0) / 1e3, cb.get("kind"), cb.get("cores"), cb.get("p1_reads_per_s")))
for k, v in (c.get("other_workloads") or {}).items():
    if "error" in v: print("   other workload %-20s ERROR %s" % (k, v["error"][-200:])); continue
    print("   other workload %-20s %.3f M reads processed/s (%.3f M aligned), frac %.4f, %s; diffed %s, mismatches %s" % (k, (v.get("reads_processed_per_s") or 0) / 1e6, v["value"] / 1e6, v["roofline_frac"], v["kernel"], v.get("reads_diffed_vs_reference"), v.get("diff_mismatches")))
PY
cp $O/bench_default.json $O/bench_default_line.json
cat $S
