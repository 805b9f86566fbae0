#!/bin/bash
# Ninth (last) GPU call of round 4: the second pass with its reads spread over many wavefronts (every 16th lane takes
# reads: BT_BEST_RETRY_STRIDE) against 64 to a wavefront (stride 1), stride 64, and the call-by-call kernel at stride 16.
#   gpurun --timeout 225 -- 'bash scripts/r4/ninth.sh'
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
f=$O/big_pe_stride
timeout 170 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none --env-sweep "stride1:BT_BEST_RETRY_STRIDE=1;stride64:BT_BEST_RETRY_STRIDE=64;stride16_call_by_call:BT_BEST_RETRY_NESTED=1" > $f.json 2> $f.log
echo "== big_pe_n1_best_50, second pass over every 16th lane (default now)" | tee -a $S
grep -E "main measurement|verify|env-sweep" $f.log | sed 's/^\[bench\] /   /' | tee -a $S
python -c "import json; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); print('   reads searched again in the last step:', d['roofline'].get('reads_searched_again_last_step'), ' still flagged:', d['config'].get('reads_overflowed'))" 2>&1 | tail -1 | tee -a $S
timeout 50 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "arena_overflow_retry or config5" > $O/parity.txt 2>&1
echo "overflow-retry / config-5 GPU tests: $(tail -1 $O/parity.txt)" | tee -a $S
