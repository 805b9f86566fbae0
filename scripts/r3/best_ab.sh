#!/bin/bash
# Round 3: bt_best_kernel with the front branch's record read in one go (four 16-byte loads) -- parity subset, then the
# two hg19-scale best-first workloads against the rates of the final call (3.09 / 0.845 M reads/s).
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best_first_vs_oracle or best_first_arena or best_first_large or paired_vs_oracle or config5 or (paired_matches and n1) or (without_best and n2_X500)" > $O/parity.txt 2>&1; say "best-first / paired parity subset: $(tail -1 $O/parity.txt)"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']))" 2>&1 | tail -1; }
f=$O/bench_big_pe; timeout 240 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 12.5 M pairs (final call: 3.091 M reads/s, 8087 ms/step): $(val $f.json)"
f=$O/bench_big_n2_best; timeout 300 python bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 16 M reads (final call: 0.845 M reads/s at 32 M): $(val $f.json)"
for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do f=$O/bench_$wl; timeout 120 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$wl (r3g, six waves: 7.572 / 28.007): $(val $f.json)"; done
cat $S
