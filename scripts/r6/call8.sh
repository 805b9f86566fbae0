#!/bin/bash
# GPU call 8 of round 6: (A) the best-first defaults after calls 6-7 (pairs: cold 40, take 24, arenas of 32 K words; single reads: take
# 8); (B) carry-over between 200 M-read steps when there are enough steps to pay for the closing launch -- the driver times 20 steps
# (round 5 measured it over 2 steps + the closing launch: 14.6 against 15.3 M reads/s, and left it off for steps this large).
#   gpurun --timeout 2400 -- 'bash scripts/r6/call8.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_8; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
line() { python - "$1" "$2" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3f M reads processed/s (%.3f M aligned), %.1f ms/step over %d steps, kernel %s avg %.1f ms, frac %.4f, verified %s, carry-over %s, mean active lanes %.1f" % (
        sys.argv[2], d["reads_processed_per_s"] / 1e6, d["value"] / 1e6, d["ms_per_step"], d["steps"], r["kernel"], r["kernel_ms_avg"], r["frac"],
        d["config"].get("hits_verified_against_text"), r.get("carry_over_launches"), r.get("mean_active_lanes_per_round", 0)))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
}
for wl in big_pe_n1_best_50 big_n2_best_100 big_pe_n1_50_v1; do
	timeout 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --also none > $O/$wl.json 2> $O/$wl.log
	line $O/$wl.json "$wl x 2 steps, the tree's defaults"
done
BT_BEST_ARENA_WORDS=65536 timeout 600 python bench.py --workload big_pe_n1_50_v1 --steps 2 --warmup 1 --no-cpu --also none > $O/v1_64k.json 2> $O/v1_64k.log
line $O/v1_64k.json "big_pe_n1_50_v1 x 2 steps, arenas of 64 K words"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu --also none > $O/search_nocarry.json 2> $O/search_nocarry.log
line $O/search_nocarry.json "big_n2_100 200 M x 10 steps (2 warm-up), no carry-over (the default)"
timeout 900 python bench.py --steps 10 --warmup 2 --carry 12 --no-cpu --also none > $O/search_carry12.json 2> $O/search_carry12.log
line $O/search_carry12.json "big_n2_100 200 M x 10 steps (2 warm-up), --carry 12"
timeout 900 python bench.py --steps 10 --warmup 2 --carry 2 --no-cpu --also none > $O/search_carry2.json 2> $O/search_carry2.log
line $O/search_carry2.json "big_n2_100 200 M x 10 steps (2 warm-up), --carry 2"
cat $S
