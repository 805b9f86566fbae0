#!/bin/bash
# GPU call 3 of round 6.
#  A. the locus image built a chain per sampled row (bt_loc_chain): the GPU image against the host build (tests), build seconds
#     against round 5's walk per row (BT_LOC_BUILD=rows)
#  B. bt_search_kernel: the tree (second quality level tallied: BT_L2_TALLY) against libbowtie_amd_nol2.so
#  C. bowtie-amd 192 M reads file -> /dev/null with one and with two streamed contexts per GPU (BT_CLI_STREAMS); SAM md5 of the
#     first 16 M reads, header's @PG line (the command line) left out: one stream, two streams, --no-stream
#   gpurun --timeout 2700 -- 'bash scripts/r6/call3.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_3; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
line() { python - "$1" "$2" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3f M reads processed/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s, rounds/read %.1f, rescans/read %.2f, fetches/read %.1f, locus image built in %.2f s" % (
        sys.argv[2], d["reads_processed_per_s"] / 1e6, d["value"] / 1e6, d["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["frac"],
        d["config"].get("hits_verified_against_text"), r.get("lane_iters_per_read", 0), r["ops_per_read"]["rescans"], r["ops_per_read"]["fetches"], r.get("locus_image_build_s", 0)))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "locus or probe or fresh_context or scratch_overflow or ragged or e_coli_synthetic" > $O/locus_tests.txt 2>&1
say "locus-image / probe / ragged / synthetic parity tests on the chain-built image: $(tail -1 $O/locus_tests.txt)"
grep -h "^FAILED" $O/locus_tests.txt | head -5 | tee -a $S
for lib in libbowtie_amd.so libbowtie_amd_nol2.so; do
	BT_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --also none > $O/search_${lib%.so}.json 2> $O/search_${lib%.so}.log
	line $O/search_${lib%.so}.json "big_n2_100 200 M reads x 3 steps, $lib"
done
BT_LOC_BUILD=rows timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu --no-verify --also none > $O/search_rows_build.json 2> $O/search_rows_build.log
line $O/search_rows_build.json "big_n2_100 1 step, locus image built a walk per row (round 5)"
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref --extra "--batch 8388608" > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file, --batch 8 M, one streamed context (call 2: 15.48 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
    print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	rm -f /tmp/cli_ours.sam
	for mode in "streams1" "streams2" "nostream"; do
		extra=""; st=1
		[ $mode = streams2 ] && st=2
		[ $mode = nostream ] && extra="--no-stream"
		BT_CLI_STREAMS=$st timeout 300 bowtie_amd/bowtie-amd -p 64 -S -n 2 -u 16000000 $extra -x $BASE $FQ /tmp/cli_md5.sam 2> $O/cli_md5_$mode.err
		say "SAM md5 (without the @PG line), first 16 M reads, $mode: $(grep -v '^@PG' /tmp/cli_md5.sam | md5sum | cut -d' ' -f1)  ($(grep -vc '^@' /tmp/cli_md5.sam) records)"
	done
	rm -f /tmp/cli_md5.sam
	for st in 1 2 2; do
		t0=$(date +%s.%N)
		BT_CLI_STREAMS=$st BT_VERBOSE=1 BT_IO_PROFILE=1 BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch 8388608 -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_s$st.err
		t1=$(date +%s.%N)
		python - "$t0" "$t1" "$st" "$O/cli_192m_s$st.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
print("bowtie-amd 192 M reads (the 64 M-read file three times; SAM to /dev/null), --batch 8 M, %s streamed context(s) (call 2, one: 32.65 s = 5.88 M reads/s): %.2f s = %.2f M reads/s" % (sys.argv[3], t, 192.0 / t))
err = open(sys.argv[4], errors="replace").read().splitlines()
print("\n".join("   " + l for l in err if "Stage busy" in l or "Time" in l or "at least one" in l or "locus image" in l))
tl = [l for l in err if "results back" in l]
sub = [l for l in err if "search: submitted" in l]
print("   first batch submitted: %s; first results back: %s; last: %s; end: %s" % (sub[0].split()[1] if sub else "?", tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?", [l for l in err if "  end" in l][-1].split()[1] if [l for l in err if "  end" in l] else "?"))
PY
		grep "\[timeline\]" $O/cli_192m_s$st.err > $O/cli_192m_s${st}_timeline.txt
	done
fi
cat $S
