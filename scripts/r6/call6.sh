#!/bin/bash
# GPU call 6 of round 6 (after the validation of main): the gates of bt_best_kernel's loop at three blocks per CU with the leaf in
# LDS (they were set in round 4 for four blocks and everything in scratch), and bowtie-amd's formatter threads / batch size.
#   gpurun --timeout 2400 -- 'bash scripts/r6/call6.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_6; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
SWEEP="cold8:BT_BEST_COLD_MIN=8;cold24:BT_BEST_COLD_MIN=24;cold32:BT_BEST_COLD_MIN=32;take8:BT_BEST_TAKE_MIN=8;take32:BT_BEST_TAKE_MIN=32;send2:BT_BEST_SEND_PERIOD=2;send8:BT_BEST_SEND_PERIOD=8;sendmin12:BT_BEST_SEND_MIN=12;sendmin40:BT_BEST_SEND_MIN=40;twice:BT_BEST_SWEEP_TWICE=1;arena32k:BT_BEST_ARENA_WORDS=32768;cold24take32:BT_BEST_COLD_MIN=24,BT_BEST_TAKE_MIN=32"
for wl in big_pe_n1_best_50 big_n2_best_100; do
	timeout 1200 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --no-verify --also none --env-sweep "$SWEEP" > $O/sweep_$wl.json 2> $O/sweep_$wl.log
	python - "$O/sweep_$wl.json" "$wl" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s, the tree's gates (cold 16, take 16, send every 4th round or 24 lanes): %.3f M reads processed/s, kernel %.1f ms" % (sys.argv[2], d["reads_processed_per_s"] / 1e6, d["roofline"]["kernel_ms_avg"]))
    for r in d.get("env_sweep", []):
        print("   %-14s %-50s %8.3f M reads/s  kernel %9.1f ms  same hit count: %s" % (r["label"], r["env"], r["reads_processed_per_s"] / 1e6, r["kernel_ms_avg"], r["n_hits_sum_equal"]))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
done
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref --extra "--batch 8388608" > $O/cli_64m.json 2> $O/cli_64m.err
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	rm -f /tmp/cli_ours.sam
	for cfg in "16777216 64" "16777216 128" "16777216 256" "12582912 128" "8388608 256"; do
		set -- $cfg
		t0=$(date +%s.%N)
		BT_CLI_FORMAT_THREADS=$2 BT_IO_PROFILE=1 BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch $1 -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_b$1_f$2.err
		t1=$(date +%s.%N)
		python - "$t0" "$t1" "$1" "$2" "$O/cli_192m_b$1_f$2.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
err = open(sys.argv[5], errors="replace").read().splitlines()
fm = [l for l in err if l.startswith("[io] batch of")]
tl = [l for l in err if "results back" in l]
sub = [l for l in err if "search: submitted" in l]
end = [l for l in err if l.rstrip().endswith(" end")]
print("bowtie-amd 192 M reads -> /dev/null, --batch %s, %s formatter threads: %.2f s = %.2f M reads/s; first submitted %s, first results %s, last %s, end %s; formatter: %s" % (
    sys.argv[3], sys.argv[4], t, 192.0 / t, sub[0].split()[1] if sub else "?", tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?", end[-1].split()[1] if end else "?", fm[len(fm) // 2][5:] if fm else "?"))
PY
	done
fi
cat $S
