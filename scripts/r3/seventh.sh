#!/bin/bash
# Round 3, seventh GPU call (short): why big_n2_best_100 got slow -- the on-stream second pass for reads that outgrow
# their 64 KB arena against larger arenas and against two blocks per CU; the new binary-level tests.
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
v() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, kernel %.1f ms/step, searched again %s, overflowed %s' % (d['value']/1e6, r.get('kernel_ms_avg', 0), r.get('reads_searched_again_last_step'), d['config'].get('reads_overflowed')))" 2>&1 | tail -1; }
run() { local tag=$1; shift; f=$O/bench_$tag; env "$@" timeout 240 python bench.py --workload big_n2_best_100 --reads 1000000 --steps 1 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 1 M reads, $tag: $(v $f.json)"; }
run default BT_X=0
run arena64k BT_BEST_ARENA_WORDS=65536
run lanes2percu BT_BEST_BLOCKS_PER_CU=2
run best2lib BT_LIB=libbowtie_amd_best2.so
timeout 300 python -m pytest tests/test_simple_cases.py tests/test_zz_gpu_fuzz.py -m gpu -q -x -k "every_kernel_instance or more_alignments or drops_the_pair" > $O/newtests.txt 2>&1; say "new binary-level tests: $(tail -1 $O/newtests.txt)"
cat $S
