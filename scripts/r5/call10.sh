#!/bin/bash
# Tenth GPU call of round 5: after calls 8 and 9 lost one or two of the binary's simple_tests cases per run (empty output,
# another case each time; six workers) the deferred refill is out again.  (1) the binary's cases three times under six
# workers, (2) bowtie-amd end to end on 64 M reads, (3) the default workload once more.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_10; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; rounds/read %.1f, locus %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r['lane_iters_per_read'], r.get('locus_mode')))" 2>&1 | tail -1; }
for k in 1 2 3; do
	timeout 300 python -m pytest tests/test_simple_cases.py tests/test_gpu_cli.py -m gpu -q > $O/cli_cases_$k.txt 2>&1
	say "the binary's GPU tests (simple_tests cases + test_gpu_cli), run $k, six workers: $(tail -1 $O/cli_cases_$k.txt)"
	grep -h "^FAILED" $O/cli_cases_$k.txt | head -4 | tee -a $S
done
BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads (call 1: 18.95 s = 3.38 M reads/s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l or "teardown" in l or l.rstrip().endswith(" end") or "index" in l.lower()))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
f=$O/full_200m; timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step (the default command, no CPU leg): $(val $f.json)"
cat $S
