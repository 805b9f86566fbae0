#!/bin/bash
# GPU call 16 of round 6: the copy stream on a hardware queue of its own (a stream of the greatest priority: bt_api.cpp).  Call 14/15's
# 640 M-read runs stalled 2-3 s every few batches whether the result arrays were page-locked or not: a collected batch's copies waited
# for the launch that was running, on the same hardware queue, and the next submission behind them.
#   gpurun --timeout 1200 -- 'bash scripts/r6/call16.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_16; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py tests/test_zzz_gpu_stress.py -q -m gpu -k "cli or stream or carry or ticks" > $O/gpu_tests.txt 2>&1
say "pytest -m gpu -k 'cli or stream or carry or ticks': $(tail -1 $O/gpu_tests.txt)"
grep -h "^FAILED" $O/gpu_tests.txt | head -5 | tee -a $S
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (third final call: 10.52 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
L10=$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ
gaps() { grep -a "search: submitted" $1 | awk '{if (p) printf "%.2f ", $2-p; p=$2} END {print ""}'; }
run() {   # label, tag, reads (M), list, env...
	local label="$1" tag="$2" m="$3" list="$4"; shift 4
	sleep 10
	env "$@" timeout 600 python scripts/r6/cli_run.py "$label" $O/cli_$tag.err $m -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $list /dev/null >> $S; tail -1 $S
	say "   seconds between submissions: $(gaps $O/cli_$tag.err | cut -c1-330)"
	say "   $(grep -a -E '^# reads with at least' $O/cli_$tag.err)"
}
run "640 M reads, the tree (copy stream of the greatest priority)" prio 640 $L10 A=1
run "640 M reads, the same with page-locked result arrays" priopin 640 $L10 BT_CLI_PINNED_RESULTS=1
run "640 M reads, an ordinary copy stream, GPU_MAX_HW_QUEUES=8" hwq8 640 $L10 BT_COPY_STREAM_PRIORITY=0 GPU_MAX_HW_QUEUES=8
run "640 M reads, an ordinary copy stream (call 14's)" plain 640 $L10 BT_COPY_STREAM_PRIORITY=0
run "192 M reads, the tree" d192 192 $FQ,$FQ,$FQ A=1
run "192 M reads, the tree, again" d192b 192 $FQ,$FQ,$FQ A=1
cat $S
