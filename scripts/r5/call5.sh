#!/bin/bash
# Fifth GPU call of round 5: PMC of the locus-mode search kernel (issue / wait counters at 16 M reads; FETCH_SIZE and WRITE_SIZE
# at 64 M reads per launch for profiles/traffic.json), and first runs of the workloads `--also auto` now adds.
#   gpurun --timeout 1000 -- 'bash scripts/r5/call5.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_5; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms, frac %.4f, verified %s %s; diffed vs reference %s, mismatches %s; gather ceiling %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], c.get('hits_verified_against_text'), c.get('verified_unit'), c.get('reads_diffed_vs_reference'), c.get('diff_mismatches'), r.get('gather_ceiling_source')))" 2>&1 | tail -1; }
pmc() {   # tag, counters, bench args...
	local tag=$1 grp=$2; shift 2
	( cd /tmp && rocprofv3 --pmc $grp --kernel-include-regex "bt_search" --output-format csv -d $O/pmc_$tag -- python $R/bench.py "$@" --no-cpu --no-verify --also none > $O/pmc_$tag.json 2> $O/pmc_$tag.log )
	python - "$O/pmc_$tag" "$tag" >> $S <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bt_search" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("PMC[%s] %-28s per dispatch %s" % (sys.argv[2], k, ["%.4g" % x for x in v]))
PY
}
pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" --reads 16000000 --carry 12 --steps 1 --warmup 1
pmc sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" --reads 16000000 --carry 12 --steps 1 --warmup 1
pmc fetch64 "FETCH_SIZE" --reads 64000000 --no-carry --steps 1 --warmup 0
pmc write64 "WRITE_SIZE" --reads 64000000 --no-carry --steps 1 --warmup 0
python - "$O/pmc_fetch64.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("   64 M-read launch under the profiler: %.3f M reads/s, wave rounds %.4g, lane rounds per read %.1f, mean active lanes %.1f, algorithmic bytes per launch %.4g" % (d["reads_processed_per_s"] / 1e6, r["wave_rounds_per_launch"], r["lane_iters_per_read"], r["mean_active_lanes_per_round"], r["algorithmic_bytes_per_launch"]))
PY
for wl in "ecoli_v0_36" "big_pe_n1_50_v1 --reads 6250000" "big_n2_best_100 --reads 16000000"; do
	set -- $wl
	f=$O/also_$1; timeout 400 python bench.py --workload "$@" --steps 2 --warmup 1 --cpu-diff-only --also none > $f.json 2> $f.log
	say "$wl: $(val $f.json)"
done
cat $S
