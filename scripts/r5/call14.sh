#!/bin/bash
# Last GPU call of round 5: what the driver runs at the round's end -- the GPU suite with -x, the smoke test -- on the tree as left.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_14; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 480 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1
say "python -m pytest tests/ -x -q -m gpu: $(tail -1 $O/gpu_suite.txt)"
grep -h "^FAILED" $O/gpu_suite.txt | head -5 | tee -a $S
grep -h -A25 "Error\b" $O/gpu_suite.txt | head -40 >> $S
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
say "smoke(): $(tail -1 $O/smoke.txt)"
cat $S
