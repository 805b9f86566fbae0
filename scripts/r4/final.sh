#!/bin/bash
# Final GPU call of round 4, most important first: the whole GPU suite on the final code; rocprofv3 --kernel-trace --stats of
# bench.py's default command (2 timed steps) and of BASELINE config 5's share (the automaton); bowtie-amd on 64 M reads;
# PMC traffic of bt_best_kernel at hg19 scale; the automaton compiled for eight waves per SIMD.
#   gpurun --timeout 1120 -- 'bash scripts/r4/final.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4g; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit')))" 2>&1 | tail -1; }
stats() {   # dir: the kernel rows of rocprofv3's stats
	python - "$1" <<'PY'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 6: print("   ", row)
PY
}
timeout 600 python -m pytest tests -m gpu -q -x > $O/gpu_suite.txt 2>&1
say "pytest -m gpu (whole suite, final code): $(tail -1 $O/gpu_suite.txt)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --also none > $O/bench_default.json 2> $O/bench_default.log
say "bench.py default workload under rocprofv3 --kernel-trace --stats, 2 timed steps: $(val $O/bench_default.json)"
stats $O/trace_default | tee -a $S
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_big_pe -- python $R/bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $O/bench_big_pe.json 2> $O/bench_big_pe.log
say "big_pe_n1_best_50 under rocprofv3 --kernel-trace --stats: $(val $O/bench_big_pe.json)"
stats $O/trace_big_pe | tee -a $S
cd /tmp
f=$O/bench_big_n2_best; timeout 200 python $R/bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_best_100, 16 M reads (round 3: 0.847 M reads/s): $(val $f.json)"
f=$O/bench_big_pe_best8; BT_LIB=libbowtie_amd_best8.so timeout 200 python $R/bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_pe_n1_best_50, eight waves per SIMD (libbowtie_amd_best8.so): $(val $f.json)"
cd $R
BT_LIB=libbowtie_amd_best8.so BT_BEST_NESTED=0 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best_first or paired" > $O/parity_best8.txt 2>&1
say "eight-waves build, automaton forced, best-first / paired GPU tests: $(tail -1 $O/parity_best8.txt)"
cd $R
BT_CLI_TIMELINE=1 timeout 200 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_64m.json").read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads (third call: 18.56 s = 3.45 M reads/s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l or "teardown" in l or l.rstrip().endswith(" end")))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
tail -n 12 $S
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c --kernel-include-regex bt_best --output-format csv -d $O/pmc_big_pe_$c -- python $R/bench.py --workload big_pe_n1_best_50 --steps 1 --warmup 1 --no-cpu --no-verify --also none > $O/pmc_big_pe_$c.json 2> $O/pmc_big_pe_$c.log
	python - "$O/pmc_big_pe_$c" "$c" >> $S <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bt_best" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("PMC %s of bt_best_kernel, big_pe_n1_best_50 (12.5 M pairs per launch), per dispatch: %s" % (k, ["%.4g" % x for x in v]))
PY
done
cat $S
