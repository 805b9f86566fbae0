#!/bin/bash
# GPU call 7 of round 6: the best-first loop's gates as set after call 6 (pairs: cold sweep at 32 lanes; both: streaks' ends every
# 8th round or at 40 lanes) against round 4's and a few neighbours; bowtie-amd with its new default batch (12 M on this host).
#   gpurun --timeout 2400 -- 'bash scripts/r6/call7.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_7; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
sweep() {   # workload, sweep
	timeout 1500 python bench.py --workload $1 --steps 2 --warmup 1 --no-cpu --also none --env-sweep "$2" > $O/sweep_$1.json 2> $O/sweep_$1.log
	python - "$O/sweep_$1.json" "$1" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s, the tree's gates: %.3f M reads processed/s, kernel %.1f ms, frac %.4f, verified %s" % (sys.argv[2], d["reads_processed_per_s"] / 1e6, d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["config"].get("hits_verified_against_text")))
    for r in d.get("env_sweep", []):
        print("   %-16s %-90s %8.3f M reads/s  kernel %9.1f ms  same hit count: %s" % (r["label"], r["env"], r["reads_processed_per_s"] / 1e6, r["kernel_ms_avg"], r["n_hits_sum_equal"]))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
}
sweep big_pe_n1_best_50 "round4:BT_BEST_COLD_MIN=16,BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;cold48:BT_BEST_COLD_MIN=48;cold40:BT_BEST_COLD_MIN=40;cold32arena32k:BT_BEST_ARENA_WORDS=32768;cold32arena48k:BT_BEST_ARENA_WORDS=49152;take8:BT_BEST_TAKE_MIN=8;take24:BT_BEST_TAKE_MIN=24;send16:BT_BEST_SEND_PERIOD=16;sendmin56:BT_BEST_SEND_MIN=56"
sweep big_pe_n1_50_v1 "nested0:BT_BEST_NESTED=0"
sweep big_n2_best_100 "round4:BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;cold12:BT_BEST_COLD_MIN=12;cold20:BT_BEST_COLD_MIN=20;take8:BT_BEST_TAKE_MIN=8;send16:BT_BEST_SEND_PERIOD=16;sendmin56:BT_BEST_SEND_MIN=56;take8sendmin56:BT_BEST_TAKE_MIN=8,BT_BEST_SEND_MIN=56"
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file, default batch (round 5: 16.04 s; final call, 8 M: 14.51 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	rm -f /tmp/cli_ours.sam
	for cfg in "default 12" "default 8" "default 12"; do
		set -- $cfg
		t0=$(date +%s.%N)
		BT_CLI_CARRY=$2 BT_IO_PROFILE=1 BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_c$2.err
		t1=$(date +%s.%N)
		python - "$t0" "$t1" "$1" "$2" "$O/cli_192m_c$2.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
err = open(sys.argv[5], errors="replace").read().splitlines()
fm = [l for l in err if l.startswith("[io] batch of")]
tl = [l for l in err if "results back" in l]
sub = [l for l in err if "search: submitted" in l]
end = [l for l in err if l.rstrip().endswith(" end")]
print("bowtie-amd 192 M reads -> /dev/null, batch %s (%s), carry-over %s launches: %.2f s = %.2f M reads/s; first submitted %s, first results %s, last %s, end %s" % (
    sys.argv[3], fm[0].split(":")[0][5:] if fm else "?", sys.argv[4], t, 192.0 / t, sub[0].split()[1] if sub else "?", tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?", end[-1].split()[1] if end else "?"))
PY
	done
fi
cat $S
