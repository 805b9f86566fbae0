#!/bin/bash
# GPU call 14 of round 6, on the tree of the third final call (documents only since): more of the same evidence.
#  1. the null-stream reproduction on the tree (0 reads may differ)
#  2. the whole GPU suite six-wide, twice more, no -x
#  3. bowtie-amd on a longer input -- the 64 M-read file ten times = 640 M reads, 51 batches: the stream's steady state beyond the
#     sixteen batches of every run so far (batches completing in mid-run, buffers recycled), and the rate once the closing tail
#     is a twelfth of the run instead of a quarter; the tallies must be ten times the file's
#   gpurun --timeout 2400 -- 'bash scripts/r6/call14.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_14; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 300 python scripts/r6/repro_null_stream.py > $O/repro_tree.json 2> $O/repro_tree.log
say "repro, the tree, null stream busy: exit $? -- $(python -c "
import json
d=json.loads(open('$O/repro_tree.json').read().strip().splitlines()[-1])
print('probe us', d['probe_us'], '; reads that differ', d['reads_that_differ'], [(c['case'], c.get('n_bad'), c.get('error','')[:80]) for c in d['cases'][:5]])
" 2>&1 | tail -1)"
for i in 1 2; do
	t0=$(date +%s)
	timeout 900 python -m pytest tests -m gpu -q > $O/gpu_suite_$i.txt 2>&1
	say "pytest -m gpu (whole suite, six workers, no -x), run $i, $(( $(date +%s) - t0 )) s: $(tail -1 $O/gpu_suite_$i.txt)"
	grep -h "^FAILED" $O/gpu_suite_$i.txt | head -5 | tee -a $S
done
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	L1=$FQ; L10=$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ
	sleep 10
	timeout 400 python scripts/r6/cli_run.py "64 M reads (the file once) -> /dev/null" $O/cli_64m_null.err 64 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $L1 /dev/null >> $S; tail -1 $S
	for i in 1 2; do
		sleep 10
		timeout 600 python scripts/r6/cli_run.py "640 M reads (the file ten times) -> /dev/null, run $i" $O/cli_640m_$i.err 640 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $L10 /dev/null >> $S; tail -1 $S
	done
	for f in cli_64m_null cli_640m_1 cli_640m_2; do say "   $f: $(grep -a -E '^# reads processed|^# reads with at least|^Reported' $O/$f.err | tr '\n' ';')"; done
	grep -a "timeline" $O/cli_640m_1.err | grep -E "submitted|results back|end of input" > $O/cli_640m_timeline.txt
fi
cat $S
