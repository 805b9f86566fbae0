#!/bin/bash
# Round 3, sixth GPU call: leaner frame scans + cheaper sweep guards (bt_search_kernel), packed mate pre-filter and six
# waves per SIMD (bt_best_kernel): parity subsets, then the rates.
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "probe or ragged or carry or stream or idempot or 1024 or overflow or retries or golden_sam or paired or best_first" > $O/parity.txt 2>&1; say "parity subset: $(tail -1 $O/parity.txt)"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, kernel %.1f ms, rounds/read %.1f, frac %.4f' % (d['value']/1e6, r.get('kernel_ms_avg', 0), r.get('lane_iters_per_read',0), r['frac']))" 2>&1 | tail -1; }
f=$O/bench_16M; timeout 400 python bench.py --workload big_n2_100 --reads 16000000 --steps 6 --warmup 2 --no-cpu --no-verify --also none > $f.json 2> $f.log
say "big_n2_100 16M carry-over (r3e: 9.474): $(val $f.json)"
f=$O/bench_64M; timeout 400 python bench.py --workload big_n2_100 --reads 64000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 64M no carry (r3e: 9.530): $(val $f.json)"
f=$O/bench_64M_2blocks; BT_NO_RL3=1 timeout 400 python bench.py --workload big_n2_100 --reads 64000000 --steps 2 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log
say "big_n2_100 64M no carry, two-block build with candidate caches (BT_NO_RL3=1): $(val $f.json)"
for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do
  f=$O/bench_$wl; timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
  say "$wl (r3d best6: 6.195 / 13.895): $(val $f.json)"
done
f=$O/bench_big_pe; timeout 500 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_pe_n1_best_50 12.5 M pairs (r2: 1.22 M reads/s): $(val $f.json)"
f=$O/bench_big_n2_best; timeout 500 python bench.py --workload big_n2_best_100 --reads 8000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_best_100 8 M reads (r2: 0.43 M reads/s at 32 M): $(val $f.json)"
cat $S
