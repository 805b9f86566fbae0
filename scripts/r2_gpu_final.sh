#!/bin/bash
# round-2 final GPU pass: the default bench line (with CPU baseline + SAM diff), its rocprofv3 kernel trace, the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
timeout 420 python bench.py > $O/bench_default_N1.json 2> $O/bench_default_N1.err; tail -c 1200 $O/bench_default_N1.json | head -c 600; echo
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o default -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-verify > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err )
find $O/prof -name "*kernel_stats*" | head -3; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" && cp "$f" $O/kernel_stats_default.csv
find $O/prof -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
( time timeout 540 python -m pytest tests -m gpu -q --timeout 300 -n 3 ) > $O/gputests.txt 2>&1
tail -6 $O/gputests.txt
