#!/bin/bash
# GPU call 11 of round 6: the formatter's threads no longer write to neighbouring cache lines (a piece's text header and tally are the
# thread's own while it works); the reader's window and index ask for huge pages.  The binary on 192 M reads again.
#   gpurun --timeout 1200 -- 'bash scripts/r6/call11.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_11; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu > $O/gpu_cli.txt 2>&1
say "pytest tests/test_gpu_cli.py -m gpu: $(tail -1 $O/gpu_cli.txt)"
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (round 5: 16.04 s; final call: 14.51 s; call 10: 11.96 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
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
	run "... the reader's blocks without huge pages" noiohuge 10 - BT_IO_HUGEPAGES=0
	run "... 64 formatter threads" f64 10 - BT_CLI_FORMAT_THREADS=64
	run "... 256 formatter threads" f256 10 - BT_CLI_FORMAT_THREADS=256
	run "... -p 128" p128 10 "-p 128" A=1
	run "... batches of 8 M reads" b8m 10 "--batch 8388608" A=1
	run "... defaults, a third time" default3 10 - A=1
	grep -a "timeline" $O/cli_192m_default.err > $O/cli_192m_default_timeline.txt
fi
cat $S
