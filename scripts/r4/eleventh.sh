#!/bin/bash
# Eleventh GPU call of round 4 (the last 100 seconds of the budget): the automaton with its own layer force-inlined into
# the kernel's loop (its pieces no longer take the lane's records by reference through real calls), e_coli only: forced
# automaton against the call-by-call kernel; before the change 8.63 vs 11.98 M (single-end), 23.7 vs 49.6 M (pairs).
export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do
	f=$O/inl_$wl; BT_BEST_NESTED=0 timeout 45 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --also none --env-sweep "nested:BT_BEST_NESTED=1" > $f.json 2> $f.log
	echo "== $wl, automaton forced (main) against call by call (sweep)" | tee -a $S
	grep -E "main measurement|verify|env-sweep" $f.log | sed 's/^\[bench\] /   /' | tee -a $S
done
