#!/bin/bash
# Tenth GPU call of round 4 (what was left of the budget): the best-first / paired GPU tests on the final library
# (second pass with one lane per wavefront by default).
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best or paired or retry or config5" > $O/parity_final_lib.txt 2>&1
echo "best-first / paired GPU tests, final library: $(tail -1 $O/parity_final_lib.txt)" | tee $O/SUMMARY.txt
