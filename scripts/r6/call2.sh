#!/bin/bash
# GPU call 2 of round 6: A/B lines on the hg19-scale index, and the command line end to end.
#  1. bt_search_kernel, default command shortened (3 steps of 200 M reads, no CPU leg): the tree against round 5's three-block
#     build without the current frame's candidate cache (libbowtie_amd_nocc.so = -DBT_LITE_CC=0)
#  2. bt_best_kernel, configs 5 and --best single-end: the tree (leaf state in LDS) against libbowtie_amd_leafscratch.so
#  3. bowtie-amd file -> file: 64 M reads (SAM written, md5 against --no-stream on the first 16 M), 192 M reads (SAM to /dev/null),
#     BT_CLI_TIMELINE + BT_IO_PROFILE: the reader reads with several threads and walks a newline index now
#  4. tests/test_zz_wide_gpu.py -v (the record VERDICT r5 asked for)
#   gpurun --timeout 2700 -- 'bash scripts/r6/call2.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_2; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
line() { python - "$1" "$2" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3f M reads processed/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s, rounds/read %s" % (
        sys.argv[2], d["reads_processed_per_s"] / 1e6, d["value"] / 1e6, d["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["frac"],
        d["config"].get("hits_verified_against_text"), "%.1f" % r.get("lane_iters_per_read", 0)))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
}
for lib in libbowtie_amd.so libbowtie_amd_nocc.so libbowtie_amd.so libbowtie_amd_nocc.so; do
	k=$(ls $O | grep -c "search_${lib%.so}")
	BT_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --also none > $O/search_${lib%.so}_$k.json 2> $O/search_${lib%.so}_$k.log
	line $O/search_${lib%.so}_$k.json "big_n2_100 200 M reads x 3 steps, $lib (run $k)"
done
for wlk in big_pe_n1_best_50 big_n2_best_100; do
	for lib in libbowtie_amd.so libbowtie_amd_leafscratch.so; do
		BT_LIB=$lib timeout 600 python bench.py --workload $wlk --steps 2 --warmup 1 --no-cpu --also none > $O/${wlk}_${lib%.so}.json 2> $O/${wlk}_${lib%.so}.log
		line $O/${wlk}_${lib%.so}.json "$wlk x 2 steps, $lib"
	done
done
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref --extra "--batch 8388608" > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file, --batch 8 M (round 5: 16.04 s = 3.99 M reads/s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
    print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	for mode in "" "--no-stream"; do
		timeout 300 bowtie_amd/bowtie-amd -p 64 -S -n 2 -u 16000000 $mode -x $BASE $FQ /tmp/cli_md5.sam 2> $O/cli_md5_${mode#--}.err
		say "SAM md5, first 16 M reads, ${mode:-streamed (default)}: $(md5sum < /tmp/cli_md5.sam | cut -d' ' -f1)"
	done
	rm -f /tmp/cli_md5.sam /tmp/cli_ours.sam
	for b in 8388608 16777216; do
		t0=$(date +%s.%N)
		BT_VERBOSE=1 BT_IO_PROFILE=1 BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch $b -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_b$b.err
		t1=$(date +%s.%N)
		python - "$t0" "$t1" "$b" "$O/cli_192m_b$b.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
print("bowtie-amd 192 M reads (the 64 M-read file three times; SAM to /dev/null), --batch %s (round 5: 36.11 s = 5.32 M reads/s): %.2f s = %.2f M reads/s" % (sys.argv[3], t, 192.0 / t))
err = open(sys.argv[4], errors="replace").read().splitlines()
print("\n".join("   " + l for l in err if "Stage busy" in l or "Time" in l or "at least one" in l or "locus image" in l))
io = [l for l in err if l.startswith("[io] fastq batch")]
print("   reader, batches 2-4: " + " || ".join(l[5:] for l in io[2:5]))
tl = [l for l in err if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
		grep "\[timeline\]" $O/cli_192m_b$b.err > $O/cli_192m_b${b}_timeline.txt
	done
fi
timeout 900 python -m pytest tests/test_zz_wide_gpu.py -v -n 3 > $O/wide_gpu_tests.txt 2>&1
say "tests/test_zz_wide_gpu.py -v: $(tail -1 $O/wide_gpu_tests.txt)"
grep -h "PASSED\|FAILED\|ERROR" $O/wide_gpu_tests.txt | sed 's/^/   /' >> $S
cat $S
