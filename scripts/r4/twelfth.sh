#!/bin/bash
# Twelfth GPU call of round 4: the best-first / paired GPU tests on the final build (the automaton's layer inlined).
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best or paired or retry or config5" > $O/parity_final_build.txt 2>&1
echo "best-first / paired GPU tests, final build: $(tail -1 $O/parity_final_build.txt)" | tee $O/SUMMARY.txt
