#!/bin/bash
# GPU call 1 of round 6: the wrong-mismatch-list bug.
#  1. scripts/r6/repro_null_stream.py against the PARENT of the fix (bowtie_amd/libbowtie_amd_parent.so = commit 974cd63's
#     library, built here from a worktree): the null stream held busy while contexts are created -- expected: FAILS, every case
#  2. the same against the tree -- expected: passes;  3. the parent without the stall -- expected: passes (why nobody saw it alone)
#  4. the new regression test and the overflow tests by themselves
#  5. the whole GPU suite, six workers, no -x, three times
#   gpurun --timeout 1500 -- 'bash scripts/r6/call1.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_1; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
BT_LIB=libbowtie_amd_parent.so timeout 300 python scripts/r6/repro_null_stream.py > $O/repro_parent.json 2> $O/repro_parent.log
say "repro, parent of the fix, null stream busy: exit $? -- $(python -c "
import json,sys
d=json.loads(open('$O/repro_parent.json').read().strip().splitlines()[-1])
print('probe us', d['probe_us'], '; reads that differ', d['reads_that_differ'], '; per case (bad, of them first-pass reads, only the mismatch list, error):', [(c['case'], c.get('n_bad'), c.get('bad_among_first_pass_reads'), c.get('only_the_mismatch_list'), c.get('error','')[:80]) for c in d['cases'][:5]])
" 2>&1 | tail -1)"
timeout 300 python scripts/r6/repro_null_stream.py > $O/repro_tree.json 2> $O/repro_tree.log
say "repro, the tree, null stream busy: exit $? -- $(python -c "
import json
d=json.loads(open('$O/repro_tree.json').read().strip().splitlines()[-1])
print('probe us', d['probe_us'], '; reads that differ', d['reads_that_differ'], [(c['case'], c.get('n_bad'), c.get('error','')[:80]) for c in d['cases'][:5]])
" 2>&1 | tail -1)"
BT_LIB=libbowtie_amd_parent.so STALL_MS=0 timeout 300 python scripts/r6/repro_null_stream.py > $O/repro_parent_nostall.json 2> $O/repro_parent_nostall.log
say "repro, parent of the fix, null stream idle: exit $? -- $(python -c "
import json
d=json.loads(open('$O/repro_parent_nostall.json').read().strip().splitlines()[-1])
print('reads that differ', d['reads_that_differ'])
" 2>&1 | tail -1)"
tail -5 $O/repro_tree.log >> $S
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 0 -k "fresh_context or scratch_overflow or device_retry" > $O/new_tests.txt 2>&1
say "the regression test + overflow tests alone: $(tail -1 $O/new_tests.txt)"
for i in 1 2 3; do
	t0=$(date +%s)
	timeout 600 python -m pytest tests -m gpu -q > $O/gpu_suite_$i.txt 2>&1
	say "pytest -m gpu (whole suite, six workers, no -x), run $i, $(( $(date +%s) - t0 )) s: $(tail -1 $O/gpu_suite_$i.txt)"
	grep -h "^FAILED" $O/gpu_suite_$i.txt | head -8 | tee -a $S
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
say "smoke(): $(tail -1 $O/smoke.txt)"
cat $S
