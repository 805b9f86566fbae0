#!/bin/bash
# Round 3, last GPU call: the whole GPU suite on the final build, bench.py's new legs on the hg19-scale index, HBM traffic
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own passes) of both kernels, the binary end to end.
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
t0=$(date +%s)
timeout 840 python -m pytest tests -m gpu -q -x > $O/suite.txt 2>&1; say "GPU suite: $(tail -1 $O/suite.txt) [$(( $(date +%s) - t0 )) s wall]"
grep -E "^FAILED|^ERROR" $O/suite.txt | head -5 | tee -a $S
timeout 600 python bench.py --workload big_n2_100 --reads 16000000 --steps 3 --warmup 1 --no-cpu --also big_v2_76,big_pe_n1_best_50,big_n2_best_100 > $O/bench_also.json 2> $O/bench_also.log
python -c "
import json; d=json.loads(open('$O/bench_also.json').read().strip().splitlines()[-1])
print('big_n2_100 16M carry-over: %.2f M reads/s (aligned %.2f M), frac %.4f' % (d['value']/1e6, d['aligned_reads_per_s']/1e6, d['roofline']['frac']))
for k, v in (d['config'].get('other_workloads') or {}).items(): print(' ', k, json.dumps(v)[:600])" 2>&1 | tee -a $S
PMC_TRAFFIC=1 bash scripts/prof.sh r3i_traffic --workload big_n2_100 --steps 1 --warmup 1 --reads 16000000 --pipes 1 --carry 0 --no-verify --also none > $O/pmc_traffic.txt 2>&1
grep "PMC" $O/pmc_traffic.txt | tee -a $S
KREGEX=bt_best PMC_TRAFFIC=1 bash scripts/prof.sh r3i_traffic_best --workload ecoli_pe_n1_best_50 --steps 1 --warmup 1 --also none > $O/pmc_traffic_best.txt 2>&1
grep "PMC" $O/pmc_traffic_best.txt | tee -a $S
timeout 400 python scripts/cli_bench.py --index big --reads 32000000 --no-ref > $O/cli_streamed.json 2> $O/cli_streamed.log; say "bowtie-amd file to file, 32 M reads, default (streamed): $(python -c "import json; d=json.loads(open('$O/cli_streamed.json').read().strip().splitlines()[-1]); print('%.2f M reads/s, %.1f s; %s' % (d['bowtie_amd_reads_per_s']/1e6, d['bowtie_amd_s'], d['bowtie_amd_stderr'][-4:]))" 2>&1 | tail -1)"
timeout 200 python scripts/cli_bench.py --index big --reads 32000000 --no-ref --extra=--no-stream > $O/cli_nostream.json 2> $O/cli_nostream.log; say "the same with --no-stream: $(python -c "import json; d=json.loads(open('$O/cli_nostream.json').read().strip().splitlines()[-1]); print('%.2f M reads/s, %.1f s' % (d['bowtie_amd_reads_per_s']/1e6, d['bowtie_amd_s']))" 2>&1 | tail -1)"
cat $S
