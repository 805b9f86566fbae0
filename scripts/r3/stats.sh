#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py command (no CPU leg): the kernel's average duration next to
# bench.py's own HIP-event figure
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also none > $O/bench_under_rocprofv3.json 2> $O/bench.log
find $O/trace -name "*kernel_stats.csv" -exec head -6 {} \; | tee $O/kernel_stats_head.txt
python -c "
import json; d=json.loads(open('$O/bench_under_rocprofv3.json').read().strip().splitlines()[-1]); r=d['roofline']
print('bench.py under the profiler: %.3f M reads/s, kernel_ms_avg %.1f (HIP events, 2 timed launches), %s' % (d['value']/1e6, r['kernel_ms_avg'], r['kernel']))" | tee -a $O/kernel_stats_head.txt
