#!/bin/bash
# Sixth GPU call of round 5: one pass per call (lanes wait a round instead of going round again), text targets taken where the step is decided.
#   gpurun --timeout 900 -- 'bash scripts/r5/call6.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_6; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; rounds/read %.1f, locus %s; env_sweep %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r['lane_iters_per_read'], r.get('locus_mode'), [(e['label'], round(e['reads_processed_per_s']/1e6, 3), e['n_hits_sum_equal']) for e in d.get('env_sweep', [])]))" 2>&1 | tail -1; }
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_simple_cases.py -m gpu -q -x -k "not best and not paired and not automaton" > $O/parity_search.txt 2>&1
say "phase-program GPU tests (test_gpu_parity + simple cases, not best / paired): $(tail -1 $O/parity_search.txt)"
if grep -q "failed\|error" $O/parity_search.txt; then tail -40 $O/parity_search.txt; fi
f=$O/ab_16m; timeout 400 python bench.py --reads 16000000 --carry 12 --steps 4 --warmup 2 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 16 M reads per step, carry-over 12, 4 steps: $(val $f.json)"
f=$O/ab_16m_rowspace; BT_LOCUS_OFF=1 timeout 400 python bench.py --reads 16000000 --carry 12 --steps 4 --warmup 2 --no-cpu --no-verify --also none > $f.json 2> $f.log
say "the same in row space (BT_LOCUS_OFF=1): $(val $f.json)"
f=$O/prof_locus; BT_LIB=libbowtie_amd_prof.so timeout 280 python scripts/prof_sections.py --workload big_n2_100 --reads 16000000 --carry 12 --steps 1 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log
say "== profiling build, big_n2_100 16 M reads, locus mode"
grep "^\[prof\]" $f.log | tee -a $S
python - "$f.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("   %.3f M reads/s; lane rounds per read %.1f, wave rounds per launch %.3e, mean active lanes %.1f" % (d["reads_processed_per_s"] / 1e6, r["lane_iters_per_read"], r["wave_rounds_per_launch"], r["mean_active_lanes_per_round"]))
print("   ops per read:", json.dumps({k: round(v, 2) for k, v in r["ops_per_read"].items()}))
PY
f=$O/full_200m; timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step (the default command, no CPU leg): $(val $f.json)"
tail -2 $f.log >> $S
cat $S
