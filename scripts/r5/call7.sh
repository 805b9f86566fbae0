#!/bin/bash
# Seventh GPU call of round 5 (diagnostic): the overflow-retry test's second pass (the two-blocks-per-CU build of the kernel,
# which the twin context launches) returned the right hits with wrong mismatch lists in call 6.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_7; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
run() {  # tag, env..., then pytest -k expr as last arg
	local tag=$1; shift
	local k="${@: -1}"; set -- "${@:1:$(($#-1))}"
	env "$@" timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "$k" > $O/$tag.txt 2>&1
	say "$tag ($* ; -k '$k'): $(tail -1 $O/$tag.txt)"
}
run overflow_locus X=1 "scratch_overflow_is_retried"
run overflow_rowspace BT_LOCUS=0 "scratch_overflow_is_retried"
run norl3_locus BT_NO_RL3=1 "matches_reference_sam and multi and (syn100 or syn76) and not best"
run norl3_rowspace BT_NO_RL3=1 BT_LOCUS=0 "matches_reference_sam and multi and (syn100 or syn76) and not best"
run ragged_norl3 BT_NO_RL3=1 "vs_oracle_ragged and not best"
grep -h "AssertionError" $O/*.txt | head -10 | tee -a $S
cat $S
