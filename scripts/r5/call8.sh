#!/bin/bash
# Eighth GPU call of round 5: (1) the overflow second pass under multi-process load (scripts/r5/retry_stress.py): does call 6's
# one-off wrong mismatch list come back?  (2) the kernel with the deferred refill: parity subset, 16 M and 200 M reads per step.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_8; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; rounds/read %.1f, locus %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r['lane_iters_per_read'], r.get('locus_mode')))" 2>&1 | tail -1; }
stress() {   # name seconds env...
	local name=$1 secs=$2; shift 2
	local pids=""
	for w in 1 2 3 4 5 6; do ( env "$@" timeout $((secs + 120)) python scripts/r5/retry_stress.py --seconds $secs --tag $name.$w > $O/stress_$name.$w.json 2> $O/stress_$name.$w.err ) & pids="$pids $!"; done
	wait $pids
	python - "$name" >> $S <<PY
import json, glob, sys
name = sys.argv[1]
rounds = fails = 0
reps = []
for f in sorted(glob.glob("$O/stress_%s.*.json" % name)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print("  %s: no result (%s)" % (f, e)); continue
    rounds += d["rounds"]; fails += d["fails"]; reps += d["reports"]
print("retry stress %-12s %5d rounds x 5 cases over six processes, %d cases differed from the oracle" % (name, rounds, fails))
for r in reps[:4]:
    print("   round %d case %s: %d of %d reads differ, %d of them in the mismatch list only; %d reads went through the second pass" % (r["round"], r["case"], r["n_bad"], r["n"], r["only_the_mismatch_list"], r["retried"]))
    for x in r["first"][:1]:
        print("      read %d\n        got  %s\n        want %s" % (x["i"], x["got"][:220], x["want"][:220]))
PY
}
stress default 45 X=1
stress rowspace 30 BT_LOCUS=0
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_simple_cases.py -m gpu -q -x -k "not best and not paired and not automaton" > $O/parity_search.txt 2>&1
say "phase-program GPU tests (test_gpu_parity + simple cases, not best / paired): $(tail -1 $O/parity_search.txt)"
if grep -q "failed\|error" $O/parity_search.txt; then grep -n "FAILED\|AssertionError" $O/parity_search.txt | head -10 | tee -a $S; fi
f=$O/ab_16m; timeout 400 python bench.py --reads 16000000 --carry 12 --steps 4 --warmup 2 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 16 M reads per step, carry-over 12, 4 steps: $(val $f.json)"
f=$O/full_200m; timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step (the default command, no CPU leg): $(val $f.json)"
tail -2 $f.log >> $S
cat $S
