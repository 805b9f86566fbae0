#!/bin/bash
# GPU call 12 of round 6: the binary leaves when its output is complete (what it still holds goes with the process), four threads let
# the written batches' memory go; formatter threads again now that they do not share cache lines.
#   gpurun --timeout 1200 -- 'bash scripts/r6/call12.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_12; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu > $O/gpu_cli.txt 2>&1
say "pytest tests/test_gpu_cli.py -m gpu: $(tail -1 $O/gpu_cli.txt)"
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (round 5: 16.04 s; final call: 14.51 s; call 11: 12.16 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	run() {   # label, file tag, seconds to wait first, extra arguments ("-" = none), environment...
		local label="$1" tag="$2" wait="$3" extra="$4"; shift 4
		[ "$extra" = "-" ] && extra=""
		sleep $wait
		env "$@" timeout 400 python scripts/r6/cli_run.py "$label" $O/cli_192m_$tag.err 192 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 $extra -x $BASE $FQ,$FQ,$FQ /dev/null >> $S
		tail -1 $S
	}
	run "192 M reads -> /dev/null, the tree's defaults" default 10 - A=1
	run "... again" default2 10 - A=1
	run "... everything let go by hand (BT_CLI_TEARDOWN=1)" teardown 10 - BT_CLI_TEARDOWN=1
	run "... one reaper" reap1 10 - BT_CLI_REAPERS=1
	run "... 64 formatter threads" f64 10 - BT_CLI_FORMAT_THREADS=64
	run "... 96 formatter threads" f96 10 - BT_CLI_FORMAT_THREADS=96
	run "... 32 formatter threads" f32 10 - BT_CLI_FORMAT_THREADS=32
	run "... batches of 8 M reads" b8m 10 "--batch 8388608" A=1
	run "... batches of 8 M reads, 64 formatter threads" b8mf64 10 "--batch 8388608" BT_CLI_FORMAT_THREADS=64
	run "... defaults, a third time" default3 10 - A=1
	grep -a "timeline" $O/cli_192m_default.err > $O/cli_192m_default_timeline.txt
fi
cat $S
