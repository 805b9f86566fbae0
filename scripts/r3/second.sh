#!/bin/bash
# Round 3, second GPU call: (1) the binary's paired runs without --best on the default build (PairedBWAlignerV1 is in it
# now), (2) occupancy sweep of bt_best_kernel: the same source compiled for 2 / 3 / 4 / 6 / 8 blocks per CU (registers
# beyond the budget spilled to scratch), e_coli first, the two best on the hg19-scale index.
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 400 python -m pytest tests/test_simple_cases.py -m gpu -q -k "test_simple_case_bowtie_amd and Paired" > $O/pe_simple.txt 2>&1; say "simple_tests pairs (asis = V1, best = V2) through bowtie-amd: $(tail -1 $O/pe_simple.txt)"
timeout 400 python -m pytest tests/test_zz_gpu_fuzz.py -m gpu -q -k without_best > $O/pe_v1_fuzz.txt 2>&1; say "bowtie-amd against the live reference, pairs without --best: $(tail -1 $O/pe_v1_fuzz.txt)"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('%.3f M reads/s, kernel %.1f ms' % (d['value']/1e6, d.get('kernel_ms_avg', 0)))" 2>&1 | tail -1; }
num() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || echo 0; }
BEST=""; BESTV=0
for lib in libbowtie_amd.so libbowtie_amd_best3.so libbowtie_amd_best4.so libbowtie_amd_best6.so libbowtie_amd_best8.so; do
  tot=0
  for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do
    f=$O/bench_${wl}_${lib%.so}
    BT_LIB=$lib timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > $f.json 2> $f.log
    say "$wl $lib: $(val $f.json)"
    tot=$(python -c "print($tot + $(num $f.json))")
  done
  if python -c "import sys; sys.exit(0 if $tot > $BESTV else 1)"; then BESTV=$tot; BEST=$lib; fi
done
say "best on e_coli: $BEST"
for lib in libbowtie_amd.so $BEST; do
  for wl in big_n2_best_100 big_pe_n1_best_50; do
    f=$O/bench_${wl}_${lib%.so}
    BT_LIB=$lib timeout 500 python bench.py --workload $wl --reads 4000000 --steps 2 --warmup 1 --no-cpu > $f.json 2> $f.log
    say "$wl (4 M) $lib: $(val $f.json)"
  done
  [ "$BEST" = "libbowtie_amd.so" ] && break
done
cat $S
