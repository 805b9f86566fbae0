#!/bin/bash
# Eighth (last) GPU call of round 4: config 5's second pass (half of the step, rocprofv3) -- through the call-by-call
# kernel (new default for that pass) against the automaton, and with larger / smaller main arenas.
#   gpurun --timeout 300 -- 'bash scripts/r4/eighth.sh'
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
f=$O/big_pe_retry
timeout 280 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none --env-sweep "retry_automaton:BT_BEST_RETRY_NESTED=0;arena_512KB:BT_BEST_ARENA_WORDS=131072;arena_128KB:BT_BEST_ARENA_WORDS=32768" > $f.json 2> $f.log
echo "== big_pe_n1_best_50, second pass call by call (default now)" | tee -a $S
grep -E "main measurement|verify|env-sweep" $f.log | sed 's/^\[bench\] /   /' | tee -a $S
python -c "import json; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); print('   reads searched again in the last step:', d['roofline'].get('reads_searched_again_last_step'), ' still flagged:', d['config'].get('reads_overflowed'))" 2>&1 | tail -1 | tee -a $S
