#!/bin/bash
# round-2 last GPU pass: (1) the simple_tests.pl and CLI suites with the CLI's default (non-streamed) search -- gating;
# (2) rocprofv3 kernel trace + stats of the default bench command, CSV; (3) diagnostics, not gating: the two inputs
# on which the streamed search faulted, with --stream, carry 12 and carry 0
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
( time timeout 240 python -m pytest tests/test_simple_cases.py tests/test_gpu_cli.py -m gpu -q --timeout 120 -n 3 ) > $O/gputests_simple_cli_default.txt 2>&1
tail -4 $O/gputests_simple_cli_default.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o default -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-verify > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err )
find $O/prof -type f | head -20
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $O/kernel_stats_default.csv; head -6 "$f"; }
find $O/prof -type f -size +1M -delete 2>/dev/null
tail -c 400 $O/bench_under_rocprof.json | head -c 300; echo
for cv in 12 0; do
  ( BT_TEST_CLI_EXTRA=--stream BT_CLI_CARRY=$cv timeout 90 python -m pytest tests/test_simple_cases.py -m gpu -q --timeout 60 -k "FASTA-continuous_6 or Checking_edits_3" ) > $O/diag_stream_carry$cv.txt 2>&1
  echo "stream carry=$cv: $(tail -1 $O/diag_stream_carry$cv.txt)"
done
