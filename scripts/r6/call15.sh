#!/bin/bash
# GPU call 15 of round 6: the 640 M-read run of call 14 stalls every few batches (submissions 2-3 s apart instead of 0.77 s) when
# several batches complete together: the searcher's thread copies each one's 515 MB of results back into pageable memory before it
# submits again.  The same run with page-locked result arrays (the switch that made no difference on 16 batches).
#   gpurun --timeout 1200 -- 'bash scripts/r6/call15.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_15; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
L10=$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ
gaps() { grep -a "search: submitted" $1 | awk '{if (p) printf "%.2f ", $2-p; p=$2} END {print ""}'; }
run() {   # label, tag, reads (M), list, env...
	local label="$1" tag="$2" m="$3" list="$4"; shift 4
	sleep 10
	env "$@" timeout 600 python scripts/r6/cli_run.py "$label" $O/cli_$tag.err $m -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $list /dev/null >> $S; tail -1 $S
	say "   seconds between submissions: $(gaps $O/cli_$tag.err | cut -c1-400)"
}
run "640 M reads, page-locked result arrays" pin 640 $L10 BT_CLI_PINNED_RESULTS=1
run "640 M reads, the tree's defaults (pageable)" def 640 $L10 A=1
run "640 M reads, page-locked result arrays, again" pin2 640 $L10 BT_CLI_PINNED_RESULTS=1
run "192 M reads, page-locked result arrays" pin192 192 $FQ,$FQ,$FQ BT_CLI_PINNED_RESULTS=1
cat $S
