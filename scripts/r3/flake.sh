#!/bin/bash
# how often does test_gpu_host_batches_streamed[12] fail, alone and under load from two other GPU processes?
export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
run() { python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -p no:cacheprovider -k "host_batches_streamed" 2>&1 | tail -40 > $1; }
for i in 1 2 3 4 5 6 7 8; do run $O/alone_$i.txt; tail -1 $O/alone_$i.txt; done 2>&1 | tee $O/alone.txt
for i in 1 2 3 4; do
  (python bench.py --workload ecoli_n2_100 --steps 20 --warmup 1 --no-cpu --also none > /dev/null 2>&1 &)
  (python bench.py --workload ecoli_pe_n1_best_50 --steps 6 --warmup 1 --no-cpu --also none > /dev/null 2>&1 &)
  sleep 12
  run $O/loaded_$i.txt; tail -1 $O/loaded_$i.txt
  sleep 8
done 2>&1 | tee $O/loaded.txt
grep -h "AssertionError: streamed" -A3 $O/*.txt | head -40 | tee $O/failures.txt
