#!/bin/bash
# Fourth final GPU call of round 6 (the tree left on main after the copy stream got a hardware queue of its own; nothing
# but documents is written after it): what the driver runs at the round's end, then the profiles the bench line cites.
#  1. python -m pytest tests -x -q -m gpu (the driver's command); smoke()
#  2. python bench.py (the default command: CPU baseline, --also auto)
#  (no rocprofv3 passes: the kernels' sources are the second and third calls')
#  4. bowtie-amd 192 M reads file -> /dev/null: the defaults three times ten seconds apart, once right behind; 640 M reads once
#   gpurun --timeout 2400 -- 'bash scripts/r6/final4.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_final4; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
t0=$(date +%s)
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite_x.txt 2>&1
say "python -m pytest tests/ -x -q -m gpu (the driver's command), $(( $(date +%s) - t0 )) s: $(tail -1 $O/gpu_suite_x.txt)"
grep -h "^FAILED" $O/gpu_suite_x.txt | head -5 | tee -a $S
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
say "smoke(): $(tail -1 $O/smoke.txt)"
t0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.log
say "python bench.py (the default command), $(( $(date +%s) - t0 )) s wall:"
python - "$O/bench_default.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, c, cb = d["roofline"], d["config"], d.get("cpu_baseline", {})
    print("   value %.3f M aligned reads/s, %.3f M reads processed/s, %.1f ms/step; roofline: %s %.1f / %.0f %s = frac %.4f, kernel %s avg %.1f ms, traffic %s; gather ceiling %.0f GB/s; rounds/read %.1f; jump table %.1f GB, %.2f look-ups and %.1f steps per read; locus image %.1f GB in %.2f s" % (
        d["value"] / 1e6, d["reads_processed_per_s"] / 1e6, d["ms_per_step"], r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"], r["kernel"], r["kernel_ms_avg"], r.get("traffic"),
        r.get("gather_ceiling_GBps", 0), r.get("lane_iters_per_read", 0), r.get("jump_table_GB", 0), r.get("jump_lookups_per_read", 0), r.get("jump_steps_per_read", 0), r.get("locus_image_GB", 0), r.get("locus_image_build_s", 0)))
    print("   verified %s %s; diffed vs reference %s reads, %s mismatches; cpu_baseline %.1f k reads/s (%s, %s cores); vs_cpu_baseline %.1f" % (c.get("hits_verified_against_text"), c.get("verified_unit"), c.get("reads_diffed_vs_reference"), c.get("diff_mismatches"), cb.get("value", 0) / 1e3, cb.get("kind"), cb.get("cores"), d.get("vs_cpu_baseline", 0)))
    for k, v in (c.get("other_workloads") or {}).items():
        if "error" in v: print("   other workload %-20s ERROR %s" % (k, v["error"][-200:])); continue
        print("   other workload %-20s %.3f M reads processed/s (%.3f M aligned), frac %.4f, %s; diffed %s, mismatches %s" % (k, (v.get("reads_processed_per_s") or 0) / 1e6, v["value"] / 1e6, v["roofline_frac"], v["kernel"], v.get("reads_diffed_vs_reference"), v.get("diff_mismatches")))
except Exception as e:
    print("   FAILED to read the bench line: %s" % e)
PY
pmc() {   # tag, regex, counters, bench args...
	local tag=$1 rx=$2 grp=$3; shift 3
	( cd /tmp && rocprofv3 --pmc $grp --kernel-include-regex "$rx" --output-format csv -d $O/pmc_$tag -- python $R/bench.py "$@" --no-cpu --no-verify --also none > $O/pmc_$tag.json 2> $O/pmc_$tag.log )
	python - "$O/pmc_$tag" "$tag" "$rx" >> $S <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[3].split("|")[0] in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("   PMC[%s] %-26s per dispatch %s" % (sys.argv[2], k, ["%.4g" % x for x in v]))
PY
}


cd $R
# gpurun copies back at most 64 MiB: keep the stats, drop the raw traces and counter dumps
for d in $O/trace_default $O/pmc_*/; do
	[ -d "$d" ] || continue
	find "$d" -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_default.csv \; 2>/dev/null
	rm -rf "$d"
done
# ---- the binary ----
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (round 5: 16.04 s; first final call: 14.51 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	for i in 1 2 3; do
		sleep 10
		timeout 400 python scripts/r6/cli_run.py "192 M reads (the 64 M-read file three times) -> /dev/null, defaults, run $i" $O/cli_192m_$i.err 192 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $FQ,$FQ,$FQ /dev/null >> $S
		tail -1 $S
	done
	timeout 400 python scripts/r6/cli_run.py "... started right behind the last run" $O/cli_192m_b2b.err 192 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $FQ,$FQ,$FQ /dev/null >> $S
	sleep 10
	L10=$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ,$FQ
	timeout 600 python scripts/r6/cli_run.py "640 M reads (the file ten times) -> /dev/null, defaults" $O/cli_640m.err 640 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $L10 /dev/null >> $S
	say "   seconds between submissions: $(grep -a 'search: submitted' $O/cli_640m.err | awk '{if (p) printf "%.2f ", $2-p; p=$2} END {print ""}' | cut -c1-330)"
	say "   $(grep -a -E '^# reads with at least' $O/cli_640m.err)"
	grep -a "timeline" $O/cli_192m_1.err > $O/cli_192m_timeline.txt
	grep -a "^\[io\]" $O/cli_192m_1.err | head -40 > $O/cli_192m_io_profile.txt
fi
cat $S
