#!/bin/bash
# First GPU call of round 4: what round 3 prepared after its GPU budget was spent and could only check on the CPU.
#   gpurun --timeout 1500 -- 'bash scripts/r4/first.sh'
# 1. bt_best_kernel with -DBF_FAST_EXTEND=1 (libbowtie_amd_fastext.so; bit-identical in the host build): the best-first
#    and paired GPU tests through it, then the four best-first workloads against the default library.
# 2. The binary end to end on the hg19-scale index with BT_CLI_TIMELINE=1 (where do the stages wait?), then with
#    BT_CLI_PINNED=1 (uploads from page-locked batches), same input; outputs compared.
# 3. The rare failure of test_gpu_host_batches_streamed[12]: 20 repeats under six-process load; the test now
#    prints the reads' pool offsets when it fails.
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
# the variant libraries travel with the snapshot when they were built before the call (make -C bowtie_amd/csrc variants, ~4 min:
# do that on the CPU side, after the last source change); built here only if missing
[ -f bowtie_amd/libbowtie_amd_fastext.so ] && [ -f bowtie_amd/libbowtie_amd_fastext_ng.so ] && [ -f bowtie_amd/libbowtie_amd_fastext_ms.so ] && [ -f bowtie_amd/libbowtie_amd_fastext_nr.so ] || { make -s -C bowtie_amd/csrc variants > $O/make.txt 2>&1 || say "make variants failed: $(tail -2 $O/make.txt)"; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('%.3f M reads/s, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']))" 2>&1 | tail -1; }

# ---- 1. fast extend ----
BT_LIB=libbowtie_amd_fastext.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "best or paired or config5 or v3 or M3 or strata" > $O/fastext_parity.txt 2>&1
say "fast-extend library, best-first / paired GPU tests: $(tail -1 $O/fastext_parity.txt)"
for lib in libbowtie_amd.so libbowtie_amd_fastext.so; do
	f=$O/bench_big_pe_$lib; BT_LIB=$lib timeout 240 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_pe_n1_best_50 (round 3: 3.29 M reads/s): $(val $f.json)"
	f=$O/bench_big_n2_best_$lib; BT_LIB=$lib timeout 300 python bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_n2_best_100 16 M reads (round 3: 0.85 M reads/s): $(val $f.json)"
done
# the parts of the switch apart (_ng: no driver-level gathers, _ms: the reference's several leaf call sites, _nr: reads handed
# out a wavefront at a time, _r32 / _r4: lanes refilled when 32 / 4 wait instead of 16), on the quick workloads
for lib in libbowtie_amd.so libbowtie_amd_fastext.so libbowtie_amd_fastext_ng.so libbowtie_amd_fastext_ms.so libbowtie_amd_fastext_nr.so libbowtie_amd_fastext_r32.so libbowtie_amd_fastext_r4.so; do
	for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do f=$O/bench_${wl}_$lib; BT_LIB=$lib timeout 120 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib $wl (round 3: 6.6 / 30.8 M): $(val $f.json)"; done
done

# ---- 2. the binary, file to file ----
BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 32000000 --no-ref > $O/cli_timeline.json 2> $O/cli_timeline.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_timeline.json").read().strip().splitlines()[-1])
print("bowtie-amd 32 M reads, default: %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join(l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
PY
cp /tmp/cli_ours.sam /tmp/cli_default.sam 2>/dev/null
BT_CLI_PINNED=1 BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 32000000 --no-ref > $O/cli_pinned.json 2> $O/cli_pinned.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_pinned.json").read().strip().splitlines()[-1])
print("bowtie-amd 32 M reads, BT_CLI_PINNED=1: %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join(l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
PY
if cmp -s <(grep -v '^@PG' /tmp/cli_default.sam) <(grep -v '^@PG' /tmp/cli_ours.sam); then say "pinned run's SAM = default run's SAM"; else say "pinned run's SAM DIFFERS from the default run's"; fi

# ---- 3. the rare streamed failure ----
( for w in 1 2 3 4 5; do ( for i in 1 2 3 4; do timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "best_first_vs_oracle_ragged" > /dev/null 2>&1; done ) & done
  for i in $(seq 1 20); do timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "host_batches_streamed" > $O/streamed_$i.txt 2>&1; echo "repeat $i: $(tail -1 $O/streamed_$i.txt)"; done; wait ) > $O/streamed_repeats.txt 2>&1
say "streamed test, 20 repeats under load: $(grep -c ' passed' $O/streamed_repeats.txt) passed, $(grep -c 'failed' $O/streamed_repeats.txt) failed"
# ... and through the library that never reads a line of the pool (DESIGN.md 4.3's candidate explanation): only worth its time if
# the repeats above failed at least once
if grep -q 'failed' $O/streamed_repeats.txt && [ -f bowtie_amd/libbowtie_amd_mmsort.so ]; then
	( for w in 1 2 3 4 5; do ( for i in 1 2 3 4; do timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "best_first_vs_oracle_ragged" > /dev/null 2>&1; done ) & done
	  for i in $(seq 1 20); do BT_LIB=libbowtie_amd_mmsort.so timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "host_batches_streamed" > $O/streamed_mmsort_$i.txt 2>&1; echo "repeat $i: $(tail -1 $O/streamed_mmsort_$i.txt)"; done; wait ) > $O/streamed_mmsort_repeats.txt 2>&1
	say "the same through libbowtie_amd_mmsort.so: $(grep -c ' passed' $O/streamed_mmsort_repeats.txt) passed, $(grep -c 'failed' $O/streamed_mmsort_repeats.txt) failed"
fi
cat $S
