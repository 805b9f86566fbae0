#!/bin/bash
# Seventh GPU call of round 4: rocprofv3 showed config 5's step to be two bt_best_kernel launches of ~2 s each -- the main
# one and the second pass over the ~316 pairs that outgrow their arena, which lasts as long as its slowest pair.  With two
# contexts taking the steps in turn (bench.py --pipes 2, there since round 2) a step's second pass runs beside the next
# step's main launch.  No library change: does it pay?
#   gpurun --timeout 330 -- 'bash scripts/r4/seventh.sh'
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, verified %s %s, contexts %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), d['config'].get('pipelined_contexts')))" 2>&1 | tail -1; }
f=$O/big_pe_pipes2; timeout 150 python bench.py --workload big_pe_n1_best_50 --pipes 2 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 --pipes 2, 2 timed steps (one context: 6.17-6.22 M reads/s): $(val $f.json)"
f=$O/big_pe_pipes2_s4; timeout 100 python bench.py --workload big_pe_n1_best_50 --pipes 2 --steps 4 --warmup 2 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 --pipes 2, 4 timed steps: $(val $f.json)"
f=$O/big_n2_best_pipes2; timeout 120 python bench.py --workload big_n2_best_100 --reads 16000000 --pipes 2 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 16 M reads --pipes 2 (one context: 2.10 M reads/s): $(val $f.json)"
cat $S
