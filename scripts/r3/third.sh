#!/bin/bash
# Round 3, third GPU call: the wave-cooperative record scans (RQ_SCAN) -- probe test, parity subset, A/B against the
# previous kernel (libbowtie_amd_base.so = HEAD before the change) on the hg19-scale index at 16 M reads per launch,
# and the section timers of the profiling build.
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "probe_scan or ragged or carry or stream or idempot or 1024 or overflow or retries" > $O/parity.txt 2>&1; say "parity subset: $(tail -1 $O/parity.txt)"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s, kernel %.1f ms, rounds/read %.1f, fetches/read %.1f' % (d['value']/1e6, r.get('kernel_ms_avg', 0), r.get('lane_iters_per_read',0), r['ops_per_read']['fetches']))" 2>&1 | tail -1; }
for lib in libbowtie_amd_base.so libbowtie_amd.so; do
  f=$O/bench_16M_${lib%.so}
  BT_LIB=$lib timeout 400 python bench.py --workload big_n2_100 --reads 16000000 --steps 6 --warmup 2 --no-cpu --no-verify > $f.json 2> $f.log
  say "big_n2_100 16M carry-over $lib: $(val $f.json)"
done
for lib in libbowtie_amd_base.so libbowtie_amd.so; do
  f=$O/bench_64M_${lib%.so}
  BT_LIB=$lib timeout 400 python bench.py --workload big_n2_100 --reads 64000000 --steps 2 --warmup 1 --no-cpu > $f.json 2> $f.log
  say "big_n2_100 64M no carry $lib: $(val $f.json)"
done
BT_LIB=libbowtie_amd_prof.so timeout 400 python scripts/prof_sections.py --workload big_n2_100 --reads 16000000 --steps 1 --warmup 1 --carry 0 --no-cpu --no-verify > $O/prof.json 2> $O/prof.log
grep "\[prof\]" $O/prof.log | tee -a $S
cat $S
# ---- occupancy sweep of bt_best_kernel (second try: the variants now have the per-launch events bench.py reads) ----
num() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || echo 0; }
v2() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('%.3f M reads/s, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']))" 2>&1 | tail -1; }
BEST=""; BESTV=0
for lib in libbowtie_amd.so libbowtie_amd_best3.so libbowtie_amd_best4.so libbowtie_amd_best6.so libbowtie_amd_best8.so; do
  tot=0
  for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do
    f=$O/bench_${wl}_${lib%.so}
    BT_LIB=$lib timeout 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > $f.json 2> $f.log
    say "$wl $lib: $(v2 $f.json)"
    tot=$(python -c "print($tot + $(num $f.json))")
  done
  if python -c "import sys; sys.exit(0 if $tot > $BESTV else 1)"; then BESTV=$tot; BEST=$lib; fi
done
say "best on e_coli: $BEST"
for lib in libbowtie_amd.so $BEST; do
  f=$O/bench_big_pe_${lib%.so}
  BT_LIB=$lib timeout 300 python bench.py --workload big_pe_n1_best_50 --reads 2000000 --steps 2 --warmup 1 --no-cpu > $f.json 2> $f.log
  say "big_pe_n1_best_50 (2 M pairs) $lib: $(v2 $f.json)"
  [ "$BEST" = "libbowtie_amd.so" ] && break
done
cat $S
