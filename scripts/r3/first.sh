#!/bin/bash
# First GPU call of round 3: settles what round 2 left unmeasured.
#   here (no GPU):   make -C bowtie_amd/csrc all variants
#   then:            gpurun --timeout 1500 -- 'bash scripts/r3/first.sh'
# Everything is bounded by its own `timeout`; results under gpurun_out/r3a/ (SUMMARY.txt first).
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
B=bowtie_amd/bowtie-amd
say() { echo "$*" | tee -a $S; }

# ---- 1. the EXT kernel instances on the two '$'-row inputs -------------------------------------
python - <<'PY' > $O/build_idx.txt 2>&1
import sys
sys.path.insert(0, "tests")
from bowtie_amd import ebwt_build as EB
from test_ebwt_build import read_fa
for ref in ("ref_08", "ref_00"):
    names, seqs = read_fa("tests/golden/simple/%s.fa" % ref)
    EB.build_index(seqs, names, "/tmp/" + ref)
PY
C100="-q -v 0 -a --quiet -x /tmp/ref_08 tests/golden/simple/case100.fq"
run() { local label=$1; shift; ( env "$@" ) > $O/$label.out 2> $O/$label.err; local rc=$?
  say "$label rc=$rc out=[$(head -c 60 $O/$label.out | tr '\n\t' '| ')] $(grep -m1 -o 'Memory access fault.*' $O/$label.err | cut -c1-90)"; return $rc; }
run c100_plain            timeout 40 $B --wrapper basic-0 -p 1 $C100
run c100_forceEXT         BT_FORCE_EXT=1 timeout 40 $B --wrapper basic-0 -p 1 $C100; EXT_RC=$?
run c100_forceEXT_norl3   BT_FORCE_EXT=1 BT_NO_RL3=1 timeout 40 $B --wrapper basic-0 -p 1 $C100
run c100_stream           timeout 40 $B --wrapper basic-0 -p 1 --stream $C100; STREAM_RC=$?
if [ $EXT_RC -ne 0 ] || [ $STREAM_RC -ne 0 ]; then
  say "EXT instances still fault: per-round trace of read 0, plain against EXT (diff the two files)"
  T=bowtie_amd/libbowtie_amd_trace.so
  run c100_trace_plain    LD_PRELOAD=$T BT_TRACE_READ=0 timeout 40 $B --wrapper basic-0 -p 1 $C100
  run c100_trace_forceEXT LD_PRELOAD=$T BT_TRACE_READ=0 BT_FORCE_EXT=1 timeout 40 $B --wrapper basic-0 -p 1 $C100
  grep '^\[trace\]' $O/c100_trace_plain.err > $O/trace_plain.txt; grep '^\[trace\]' $O/c100_trace_forceEXT.err > $O/trace_ext.txt
  diff $O/trace_plain.txt $O/trace_ext.txt > $O/trace.diff; say "trace diff: $(wc -l < $O/trace.diff) lines"
else
  say "EXT instances pass on case 100: the regression test and the whole simple_tests suite through --stream"
  BT_RUN_KNOWN_FAULT=1 timeout 200 python -m pytest tests/test_simple_cases.py -m gpu -q -k ext_kernel_instances > $O/known_fault.txt 2>&1; say "known-fault test: $(tail -1 $O/known_fault.txt)"
  BT_TEST_CLI_EXTRA=--stream timeout 400 python -m pytest tests/test_simple_cases.py -m gpu -q -n 4 > $O/simple_stream.txt 2>&1; say "simple_tests via --stream: $(tail -1 $O/simple_stream.txt)"
  BT_DEVICE_RETRY=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "carry or stream or retry or gated" > $O/device_path.txt 2>&1; say "device-path tests: $(tail -1 $O/device_path.txt)"
fi

# ---- 2. paired-end without --best on the GPU: the -DBT_PE_V1 build --------------------------------
BT_LIB=libbowtie_amd_pev1.so BT_RUN_UNVERIFIED=1 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k without_best > $O/pe_v1.txt 2>&1
say "PairedBWAlignerV1 on the GPU: $(tail -1 $O/pe_v1.txt)"
# the binary on that library (it asks bt_has_pe_v1): the paired simple_tests.pl cases as written, and the fuzz against the live reference
BT_LIB=libbowtie_amd_pev1.so LD_PRELOAD=$PWD/bowtie_amd/libbowtie_amd_pev1.so BT_SIMPLE_PAIRED_VARIANT=asis timeout 300 python -m pytest tests/test_simple_cases.py -m gpu -q -k test_simple_case_bowtie_amd > $O/pe_v1_simple.txt 2>&1
say "  simple_tests.pl pairs without --best through bowtie-amd: $(tail -1 $O/pe_v1_simple.txt)"
BT_LIB=libbowtie_amd_pev1.so LD_PRELOAD=$PWD/bowtie_amd/libbowtie_amd_pev1.so timeout 300 python -m pytest tests/test_zz_gpu_fuzz.py -m gpu -q -k without_best > $O/pe_v1_fuzz.txt 2>&1
say "  bowtie-amd against the live reference, pairs without --best: $(tail -1 $O/pe_v1_fuzz.txt)"

# ---- 3. bt_best_kernel at two blocks per CU (256 VGPRs, no spill) against the default (266, one block) ---------
for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do
  for lib in libbowtie_amd.so libbowtie_amd_best2.so; do
    BT_LIB=$lib timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > $O/bench_${wl}_${lib%.so}.json 2> $O/bench_${wl}_${lib%.so}.log
    say "$wl $lib: $(python -c "import json,sys; d=json.loads(open('$O/bench_${wl}_${lib%.so}.json').read().strip().splitlines()[-1]); print('%.3f M reads/s, kernel %.1f ms' % (d['value']/1e6, d.get('kernel_ms_avg', 0)))" 2>&1 | tail -1)"
  done
done
[ -n "$R3_BEST2_TESTS" ] && BT_LIB=libbowtie_amd_best2.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "best or paired" > $O/best2_tests.txt 2>&1; say "best2 build, best-first + paired GPU tests: $(tail -1 $O/best2_tests.txt)"
cat $S
