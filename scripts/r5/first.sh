#!/bin/bash
# First GPU call proposed for round 5 (written at the end of round 4, whose GPU budget was spent; nothing in it changes the
# product).  Before the call, on the CPU side:   make -C bowtie_amd/csrc bestprof bestsweep
#   gpurun --timeout 700 -- 'bash scripts/r5/first.sh'
# 1. Where does the automaton's wavefront time go?  Section timers of its own pieces (HOT / HOT_STEP / HOT_SEND / HOT_CHASE,
#    COLD / COLD_TAKE / COLD_EXIT / COLD_POST / COLD_RUN / COLD_PRE) on the two hg19-scale best-first workloads.
# 2. Its state lives in scratch (1 248 B per lane x 262 144 lanes = 327 MB: more than L2 + Infinity Cache, and PMC shows
#    52 KB *written* per read): the same source with a larger register budget -- two / three waves per SIMD (256 / 168
#    registers) -- against the product's four.  (Eight waves were measured in round 4: slower.)
# 3. Two contexts in turn (--pipes 2) now that the second pass is short.
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, %.1f ms/step, kernel %s avg %.1f ms, verified %s %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit')))" 2>&1 | tail -1; }
for wl in "big_pe_n1_best_50 --reads 4000000" "big_n2_best_100 --reads 4000000"; do
	set -- $wl
	f=$O/prof_$1; BT_LIB=libbowtie_amd_bestprof.so timeout 200 python bench.py --workload "$@" --steps 1 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log
	say "== section profile, $wl"; grep "section" $f.log | sed 's/^\[bench\] //' | tee -a $S
done
for lib in libbowtie_amd.so libbowtie_amd_best3.so libbowtie_amd_best2.so; do
	f=$O/occ_big_pe_$lib; BT_LIB=$lib timeout 150 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 $lib: $(val $f.json)"
	f=$O/occ_big_n2_best_$lib; BT_LIB=$lib timeout 150 python bench.py --workload big_n2_best_100 --reads 8000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 8 M reads $lib: $(val $f.json)"
done
f=$O/big_pe_pipes2; timeout 150 python bench.py --workload big_pe_n1_best_50 --pipes 2 --steps 4 --warmup 2 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 --pipes 2, 4 steps: $(val $f.json)"
cat $S
