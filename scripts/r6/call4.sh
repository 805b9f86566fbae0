#!/bin/bash
# GPU call 4 of round 6.
#  A. parity on the tree: the jump-table tests, the fewer-arenas test, locus / ragged / synthetic / overflow / fresh-context tests
#  B. bt_search_kernel (200 M reads x 3 steps): the tree (candidate cache + jump table, LDS 53 200 B = 3 blocks per CU again) with
#     and without the jump table (BT_JUMP=0), and the build with the second quality level instead of the candidate cache
#  C. bt_best_kernel at three blocks per CU (168 registers, leaf state in LDS) against the tree's four
#  D. bowtie-amd 192 M reads file -> /dev/null (raw-buffer SAM writer, 128 formatter threads, no runtime teardown)
#   gpurun --timeout 2700 -- 'bash scripts/r6/call4.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_4; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
line() { python - "$1" "$2" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3f M reads processed/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s, rounds/read %.1f, fetches/read %.1f, locus image %.2f s" % (
        sys.argv[2], d["reads_processed_per_s"] / 1e6, d["value"] / 1e6, d["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["frac"],
        d["config"].get("hits_verified_against_text"), r.get("lane_iters_per_read", 0), r["ops_per_read"]["fetches"], r.get("locus_image_build_s", 0)))
except Exception as e:
    print("%s: FAILED (%s)" % (sys.argv[2], e))
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "jump_table or fewer_arenas or locus or probe or fresh_context or scratch_overflow or ragged or e_coli_synthetic or matches_reference_sam" > $O/tests.txt 2>&1
say "jump-table / fewer-arenas / locus / ragged / synthetic / golden SAM tests: $(tail -1 $O/tests.txt)"
grep -h "^FAILED" $O/tests.txt | head -8 | tee -a $S
grep -h -B2 -A12 "Error" $O/tests.txt | head -60 >> $O/tests_errors.txt
BT_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --also none > $O/search_tree.json 2> $O/search_tree.log
line $O/search_tree.json "big_n2_100 200 M x 3, the tree (candidate cache, jump table of 14 characters)"
BT_JUMP=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --also none > $O/search_nojump.json 2> $O/search_nojump.log
line $O/search_nojump.json "big_n2_100 200 M x 3, BT_JUMP=0"
BT_LIB=libbowtie_amd_l2.so BT_JUMP=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --also none > $O/search_l2.json 2> $O/search_l2.log
line $O/search_l2.json "big_n2_100 200 M x 3, second quality level tallied instead of the candidate cache (libbowtie_amd_l2.so), BT_JUMP=0"
timeout 600 python bench.py --workload big_v2_76 --steps 3 --warmup 1 --no-cpu --also none > $O/v2_tree.json 2> $O/v2_tree.log
line $O/v2_tree.json "big_v2_76 50 M x 3, the tree"
BT_JUMP=0 timeout 600 python bench.py --workload big_v2_76 --steps 3 --warmup 1 --no-cpu --also none > $O/v2_nojump.json 2> $O/v2_nojump.log
line $O/v2_nojump.json "big_v2_76 50 M x 3, BT_JUMP=0"
for wlk in big_pe_n1_best_50 big_n2_best_100; do
	for lib in libbowtie_amd.so libbowtie_amd_best3.so; do
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
    print("bowtie-amd 64 M reads file -> SAM file, --batch 8 M (call 3: 15.51 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
    print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	rm -f /tmp/cli_ours.sam
	for mode in "stream" "nostream"; do
		extra=""; [ $mode = nostream ] && extra="--no-stream"
		timeout 300 bowtie_amd/bowtie-amd -p 64 -S -n 2 -u 16000000 $extra -x $BASE $FQ /tmp/cli_md5.sam 2> $O/cli_md5_$mode.err
		say "SAM md5 (without the @PG line), first 16 M reads, $mode: $(grep -v '^@PG' /tmp/cli_md5.sam | md5sum | cut -d' ' -f1)  ($(grep -vc '^@' /tmp/cli_md5.sam) records; call 3: 13aea54799f2843d01f4f25292ba4e81)"
	done
	rm -f /tmp/cli_md5.sam
	for rep in 1 2; do
		t0=$(date +%s.%N)
		BT_VERBOSE=1 BT_IO_PROFILE=1 BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch 8388608 -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_$rep.err
		t1=$(date +%s.%N)
		python - "$t0" "$t1" "$rep" "$O/cli_192m_$rep.err" >> $S <<'PY'
import sys, re
t = float(sys.argv[2]) - float(sys.argv[1])
print("bowtie-amd 192 M reads (the 64 M-read file three times; SAM to /dev/null), --batch 8 M, run %s (call 3: 27.83 s = 6.90 M reads/s; round 5: 36.11 s): %.2f s = %.2f M reads/s" % (sys.argv[3], t, 192.0 / t))
err = open(sys.argv[4], errors="replace").read().splitlines()
print("\n".join("   " + l for l in err if "Stage busy" in l or "Time" in l or "at least one" in l or "locus image" in l))
io = [l for l in err if l.startswith("[io] fastq batch")]
tot = []
for l in io[2:12]:
    m = re.findall(r"(\d+\.\d+)", l.split(":", 1)[1])
    # window+scan, (reading, indexing), buffers, records, names
    if len(m) >= 6: tot.append(float(m[0]) + float(m[3]) + float(m[4]) + float(m[5]))
if tot: print("   reader, batches 2-11: %.3f s per batch of 8 388 608 reads = %.1f M reads/s parsed (one line: %s)" % (sum(tot) / len(tot), 8.388608 / (sum(tot) / len(tot)), io[3][5:] if len(io) > 3 else ""))
fm = [l for l in err if l.startswith("[io] batch of")]
if fm: print("   formatter: " + fm[len(fm) // 2][5:])
tl = [l for l in err if "results back" in l]
sub = [l for l in err if "search: submitted" in l]
end = [l for l in err if l.rstrip().endswith(" end")]
print("   first batch submitted: %s; first results back: %s; last: %s; end: %s" % (sub[0].split()[1] if sub else "?", tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?", end[-1].split()[1] if end else "?"))
PY
		grep "\[timeline\]" $O/cli_192m_$rep.err > $O/cli_192m_${rep}_timeline.txt
	done
fi
cat $S
