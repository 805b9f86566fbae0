#!/bin/bash
# GPU call 17 of round 6 (the round's last minutes; the tree of the fourth final call): the 640 M-read run twice more.
#   gpurun --timeout 600 -- 'bash scripts/r6/call17.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_17; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
BT_CLI_TIMELINE=0 timeout 400 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
L10=$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ
for i in 1 2; do
	sleep 8
	timeout 200 python scripts/r6/cli_run.py "640 M reads (the file ten times) -> /dev/null, defaults, run $i" $O/cli_640m_$i.err 640 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $L10 /dev/null >> $S
	say "   seconds between submissions: $(grep -a 'search: submitted' $O/cli_640m_$i.err | awk '{if (p) printf "%.2f ", $2-p; p=$2} END {print ""}' | cut -c1-330)"
	say "   $(grep -a -E '^# reads with at least' $O/cli_640m_$i.err)"
done
cat $S
