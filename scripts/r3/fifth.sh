#!/bin/bash
# Round 3, fifth GPU call: instruction counters (rocprofv3 --pmc, SQ groups, own passes) and section timers of the
# rank-block kernel on the hg19-scale index; bench.py's new legs (--also, strong pass over gloo) on small workloads.
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
PMC_ONLY=1 bash scripts/prof.sh r3f_pmc --workload big_n2_100 --steps 1 --warmup 1 --reads 16000000 --pipes 1 --carry 0 --no-verify --also none > $O/pmc_summary.txt 2>&1
grep "PMC" $O/pmc_summary.txt | tee -a $S
BT_LIB=libbowtie_amd_prof.so timeout 400 python scripts/prof_sections.py --workload big_n2_100 --reads 16000000 --steps 1 --warmup 1 --carry 0 --no-cpu --no-verify --also none > $O/prof.json 2> $O/prof.log
grep "\[prof\]" $O/prof.log | tee -a $S
timeout 300 python bench.py --workload ecoli_n2_100 --steps 2 --warmup 1 --also ecoli_v2_76,ecoli_pe_n1_best_50 > $O/bench_also.json 2> $O/bench_also.log
python -c "
import json; d=json.loads(open('$O/bench_also.json').read().strip().splitlines()[-1])
print('ecoli_n2_100 %.2f M reads/s aligned %.2f M; diff %s/%s; other: %s' % (d['value']/1e6, d['aligned_reads_per_s']/1e6, d['config'].get('diff_mismatches'), d['config'].get('reads_diffed_vs_reference'), json.dumps(d['config'].get('other_workloads'))[:900]))" 2>&1 | tee -a $S
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "two_ranks" > $O/two_ranks.txt 2>&1; say "two-rank gloo bench test: $(tail -1 $O/two_ranks.txt)"
cat $S
