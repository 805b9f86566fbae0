#!/bin/bash
# Third GPU call of round 5: where does a round's time go with locus mode?  (profiling build: section timers)
#   gpurun --timeout 600 -- 'bash scripts/r5/call3.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_3; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
for mode in locus rowspace; do
	if [ $mode = rowspace ]; then export BT_LOCUS_OFF=1; else unset BT_LOCUS_OFF; fi
	f=$O/prof_$mode; BT_LIB=libbowtie_amd_prof.so timeout 280 python scripts/prof_sections.py --workload big_n2_100 --reads 16000000 --carry 12 --steps 1 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log
	say "== profiling build, big_n2_100 16 M reads, $mode"
	grep "^\[prof\]" $f.log | tee -a $S
	python - "$f.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("   %.3f M reads/s; lane rounds per read %.1f, wave rounds per launch %.3e, mean active lanes %.1f" % (d["reads_processed_per_s"] / 1e6, r["lane_iters_per_read"], r["wave_rounds_per_launch"], r["mean_active_lanes_per_round"]))
print("   ops per read:", json.dumps({k: round(v, 2) for k, v in r["ops_per_read"].items()}))
PY
done
unset BT_LOCUS_OFF
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "overflow" > $O/parity_overflow.txt 2>&1
say "overflow / retry GPU tests with the smaller entry cap: $(tail -1 $O/parity_overflow.txt)"
cat $S
