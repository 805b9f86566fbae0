#!/bin/bash
# Second GPU call of round 5: locus mode on the device for the first time.
#   gpurun --timeout 900 -- 'bash scripts/r5/call2.sh'
# 1. parity: the locus image against the host build, locus mode against row space, then every phase-program GPU test.
# 2. big_n2_100 at 16 M reads per step (carry-over 12): locus mode, then row space in the same process (--env-sweep).
# 3. the default command's workload at full size (200 M reads per step), 1 warm-up + 2 timed steps, every hit verified.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_2; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; rounds/read %.1f, locus %s image %.1f GB built in %.1f s; env_sweep %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r['lane_iters_per_read'], r.get('locus_mode'), r.get('locus_image_GB', 0), r.get('locus_image_build_s', 0), [(e['label'], round(e['reads_processed_per_s']/1e6, 3), e['n_hits_sum_equal']) for e in d.get('env_sweep', [])]))" 2>&1 | tail -1; }

timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "locus" > $O/parity_locus.txt 2>&1
say "locus tests: $(tail -1 $O/parity_locus.txt)"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_simple_cases.py -m gpu -q -x -k "not best and not paired and not automaton and not locus" > $O/parity_search.txt 2>&1
say "phase-program GPU tests (test_gpu_parity + simple cases, not best / paired): $(tail -1 $O/parity_search.txt)"
if ! grep -q "passed" $O/parity_locus.txt || grep -q "failed\|error" $O/parity_locus.txt; then tail -40 $O/parity_locus.txt; fi
if grep -q "failed\|error" $O/parity_search.txt; then tail -40 $O/parity_search.txt; fi

f=$O/ab_16m; BT_VERBOSE=1 timeout 400 python bench.py --reads 16000000 --carry 12 --steps 4 --warmup 2 --no-cpu --also none --env-sweep "rowspace:BT_LOCUS_OFF=1" > $f.json 2> $f.log
say "big_n2_100 16 M reads per step, carry-over 12: $(val $f.json)"
grep "locus image\|no room" $f.log | tee -a $S
python - "$f.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   ops per read:", json.dumps({k: round(v, 2) for k, v in d["roofline"]["ops_per_read"].items()}))
PY
f=$O/full_200m; BT_VERBOSE=1 timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step (the default command, no CPU leg): $(val $f.json)"
tail -3 $f.log >> $S
cat $S
