#!/bin/bash
# GPU call 9 of round 6: bowtie-amd's host side after round 6's timeline of a 192 M-read run (profiles/r6/call7_*): large blocks
# mapped with transparent huge pages, result arrays in page-locked memory, batches let go on a thread of their own, one more batch
# in flight -- each against its switch; and a few more points for the best-first gates of pairs.
#   gpurun --timeout 2400 -- 'bash scripts/r6/call9.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_9; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
# ---- the host ----
say "host: $(nproc) hardware threads, $(awk '/MemTotal/ {printf "%.0f GB", $2/1048576}' /proc/meminfo), transparent huge pages: enabled=$(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null) defrag=$(cat /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null), kernel $(uname -r)"
gcc -O2 -fopenmp scripts/r6/touch_probe.c -o /tmp/touch_probe 2>> $O/probe.err && { OMP_NUM_THREADS=64 /tmp/touch_probe 16 0; OMP_NUM_THREADS=64 /tmp/touch_probe 16 1; } 2>&1 | sed 's/^/   16 GB on 64 threads: /' | tee -a $S
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (round 5: 16.04 s; final call: 14.51 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	run() {   # label, file tag, environment...
		local label="$1" tag="$2"; shift 2
		env "$@" timeout 400 python scripts/r6/cli_run.py "$label" $O/cli_192m_$tag.err 192 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $FQ,$FQ,$FQ /dev/null >> $S
		tail -1 $S
	}
	run "192 M reads -> /dev/null, the tree's defaults" default A=1
	run "... 14 batches in flight" fl14 BT_CLI_INFLIGHT=14
	run "... malloc()ed blocks (BT_CLI_HUGEPAGES=0)" nohuge BT_CLI_HUGEPAGES=0
	run "... pageable result arrays" nopin BT_CLI_PINNED_RESULTS=0
	run "... both off (the reaper thread alone)" neither BT_CLI_HUGEPAGES=0 BT_CLI_PINNED_RESULTS=0
	run "... 14 in flight, carry-over 11" fl14c11 BT_CLI_INFLIGHT=14 BT_CLI_CARRY=11
	run "... 14 in flight, read batches page-locked too" fl14pin BT_CLI_INFLIGHT=14 BT_CLI_PINNED=1
	run "... 14 in flight, 256 formatter threads" fl14f256 BT_CLI_INFLIGHT=14 BT_CLI_FORMAT_THREADS=256
	run "... the tree's defaults again" default2 A=1
	# the text does not depend on any of it: 4 M reads to a file, with and without
	head -n 16000000 $FQ > /tmp/r4m.fq
	for cfg in "A=1" "BT_CLI_HUGEPAGES=0 BT_CLI_PINNED_RESULTS=0" "BT_CLI_INFLIGHT=14 BT_CLI_BIG_MIN=65536"; do
		env $cfg bowtie_amd/bowtie-amd -p 64 -S -n 2 --batch 500000 -x $BASE /tmp/r4m.fq /tmp/r4m.sam 2> $O/r4m.err
		say "4 M reads in batches of 500 000 -> SAM file, $cfg: md5 without @PG $(grep -v '^@PG' /tmp/r4m.sam | md5sum | cut -c1-32), $(grep -vc '^@' /tmp/r4m.sam) records"
	done
	rm -f /tmp/r4m.sam /tmp/r4m.fq
fi
# ---- the best-first gates of pairs: a few more points around the defaults ----
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
sweep big_pe_n1_best_50 "arena24k:BT_BEST_ARENA_WORDS=24576;arena16k:BT_BEST_ARENA_WORDS=16384;cold48:BT_BEST_COLD_MIN=48;take32:BT_BEST_TAKE_MIN=32;sendmin48:BT_BEST_SEND_MIN=48;send12:BT_BEST_SEND_PERIOD=12;arena24kcold48:BT_BEST_ARENA_WORDS=24576,BT_BEST_COLD_MIN=48"
sweep big_n2_best_100 "arena96k:BT_BEST_ARENA_WORDS=98304;take4:BT_BEST_TAKE_MIN=4;cold16send12:BT_BEST_SEND_PERIOD=12"
cat $S
