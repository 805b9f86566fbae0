#!/bin/bash
# Sixth GPU call of round 4: the wavefront automaton (default gates 16/16/4/24, no second sweep pass) against the
# call-by-call kernel, hg19 scale first; the fifth call lost its numbers to a gate setting that could not make progress
# (fixed: lanes that wait for a read no longer count towards a sweep that cannot give them one).
#   gpurun --timeout 480 -- 'bash scripts/r4/sixth.sh'
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
run() {   # seconds workload sweep extra...
	local secs=$1 wl=$2 sw=$3; shift 3
	local f=$O/ab_$wl
	timeout $secs python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --also none --env-sweep "$sw" "$@" > $f.json 2> $f.log
	say "== $wl $*"
	grep -E "main measurement|verify|env-sweep" $f.log | sed 's/^\[bench\] /   /' | tee -a $S
}
run 200 big_pe_n1_best_50 "nested:BT_BEST_NESTED=1;s1:BT_BEST_SEND_PERIOD=1,BT_BEST_SEND_MIN=8;c32:BT_BEST_COLD_MIN=32"
run 220 big_n2_best_100 "nested:BT_BEST_NESTED=1;s1:BT_BEST_SEND_PERIOD=1,BT_BEST_SEND_MIN=8;c32:BT_BEST_COLD_MIN=32" --reads 8000000
SW_E="nested:BT_BEST_NESTED=1;s1:BT_BEST_SEND_PERIOD=1,BT_BEST_SEND_MIN=8;s8:BT_BEST_SEND_PERIOD=8,BT_BEST_SEND_MIN=32;c32:BT_BEST_COLD_MIN=32;c8:BT_BEST_COLD_MIN=8;t32:BT_BEST_TAKE_MIN=32;twice:BT_BEST_SWEEP_TWICE=1"
run 90 ecoli_n2_best_100 "$SW_E"
run 90 ecoli_pe_n1_best_50 "$SW_E"
cat $S
