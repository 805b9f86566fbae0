#!/bin/bash
# Ninth GPU call of round 5: descriptors go to the device from page-locked slots (ctx_h2d); the best-first engine resolves a
# reported row through the dense suffix array.  (1) the phase-program parity subset twice under six xdist workers (the load
# under which calls 6 and 8 each lost a test), (2) the retry stress again, (3) best-first / paired parity + config 5's share
# and --best single-end, (4) the search kernel at 16 M and 200 M reads per step.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_9; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; rounds/read %.1f, locus %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r['lane_iters_per_read'], r.get('locus_mode')))" 2>&1 | tail -1; }
for k in 1 2; do
	timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_simple_cases.py -m gpu -q -k "not best and not paired and not automaton" > $O/parity_search_$k.txt 2>&1
	say "phase-program GPU tests, run $k (six workers): $(tail -1 $O/parity_search_$k.txt)"
	grep -h "^FAILED" $O/parity_search_$k.txt | head -5 | tee -a $S
done
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
for r in reps[:3]:
    print("   round %d case %s: %d of %d reads differ, %d of them in the mismatch list only; %d reads went through the second pass" % (r["round"], r["case"], r["n_bad"], r["n"], r["only_the_mismatch_list"], r["retried"]))
PY
}
stress rowspace 40 BT_LOCUS=0
stress default 30 X=1
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "best or paired or automaton" > $O/parity_best.txt 2>&1
say "best-first / paired / automaton GPU tests: $(tail -1 $O/parity_best.txt)"
grep -h "^FAILED" $O/parity_best.txt | head -5 | tee -a $S
f=$O/big_pe; timeout 300 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 (dense SA in the chase; call 1: 10.09 M with calls, 9.57 inlined): $(val $f.json)"
f=$O/big_pe_walk; BT_BEST_LOCUS=0 timeout 300 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log; say "   the same, walking (BT_BEST_LOCUS=0): $(val $f.json)"
f=$O/big_n2_best; timeout 300 python bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 16 M reads (call 5: 2.06 M): $(val $f.json)"
f=$O/big_pe_v1; timeout 300 python bench.py --workload big_pe_n1_50_v1 --reads 6250000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_50_v1 6.25 M pairs (call 5: 11.5 M): $(val $f.json)"
f=$O/ab_16m; timeout 400 python bench.py --reads 16000000 --carry 12 --steps 4 --warmup 2 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 16 M reads per step, carry-over 12, 4 steps: $(val $f.json)"
f=$O/full_200m; timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step (the default command, no CPU leg): $(val $f.json)"
cat $S
