#!/bin/bash
# First GPU call of round 5: measurements that decide what gets merged / built, nothing here changes the product.
#   gpurun --timeout 1000 -- 'bash scripts/r5/call1.sh'
# 1. bt_search_kernel, profiling build: how many of a read's rank rounds run on a range that is ONE BWT row (mapLF1 + the
#    mapLFEx calls whose two rows are neighbours), and how many runs of such rounds there are -- the hg19-scale numbers
#    behind "locus mode" (scripts/textmode_model.py has the e_coli / 30 Mbp ones).
# 2. bt_best_kernel A/B at hg19 scale: main  vs  -DBF_NO_FNI (round 4 before cc149e9)  vs  r5-prep's one rank point per
#    hot round.
# 3. bt_align_stream_tick: the GPU test, then bowtie-amd on 64 M reads with and without ticks (same SAM md5?).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_1; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit')))" 2>&1 | tail -1; }

f=$O/prof_single_row; BT_LIB=libbowtie_amd_prof.so timeout 300 python scripts/prof_sections.py --workload big_n2_100 --reads 16000000 --carry 12 --steps 1 --warmup 1 --no-cpu --no-verify --also none > $f.json 2> $f.log
say "== profiling build, big_n2_100 16 M reads: $(val $f.json)"
grep "^\[prof\]" $f.log | tee -a $S
python - "$f.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("op counts per read:", json.dumps(r.get("ops_per_read")), "lane rounds per read %.1f" % r.get("lane_iters_per_read", 0), "reads per launch", d["config"].get("reads_per_gpu_per_step"))
PY

for lib in libbowtie_amd.so libbowtie_amd_nofni.so libbowtie_amd_r5prep.so; do
	f=$O/best_big_pe_$lib; BT_LIB=$lib timeout 200 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50 $lib: $(val $f.json)"
done
for lib in libbowtie_amd.so libbowtie_amd_r5prep.so; do
	f=$O/best_big_n2_$lib; BT_LIB=$lib timeout 200 python bench.py --workload big_n2_best_100 --reads 8000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100 8 M reads $lib: $(val $f.json)"
done
BT_LIB=libbowtie_amd_r5prep.so BT_BEST_NESTED=0 timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best_first or paired" > $O/parity_r5prep.txt 2>&1
say "r5-prep library, automaton forced, best-first / paired GPU tests: $(tail -1 $O/parity_r5prep.txt)"

timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "streamed" > $O/parity_streamed.txt 2>&1
say "streamed GPU tests (incl. finished_by_ticks): $(tail -1 $O/parity_streamed.txt)"
for t in ticks noticks; do
	if [ $t = noticks ]; then export BT_CLI_NO_TICKS=1; else unset BT_CLI_NO_TICKS; fi
	BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m_$t.json 2> $O/cli_64m_$t.err
	md5sum /tmp/cli_ours.sam | cut -c1-32 > $O/cli_64m_$t.md5
	python - "$O/cli_64m_$t.json" "$t" "$(cat $O/cli_64m_$t.md5)" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads, %s (round 4: 19.0 s = 3.36 M reads/s): %.2f s = %.2f M reads/s, SAM md5 %s" % (sys.argv[2], d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6, sys.argv[3]))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l or "teardown" in l or l.rstrip().endswith(" end")))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
done
unset BT_CLI_NO_TICKS
cat $S
