#!/bin/bash
# rocprofv3 passes over one bench configuration (run on the GPU box, from the repo root).
#   scripts/prof.sh <tag> <bench args...>
# kernel-trace/stats and each PMC group run separately (never combined with sys/hip/hsa traces).
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "${PMC_ONLY:-}${SKIP_TRACE:-}${PMC_TRAFFIC:-}" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu > $OUT/trace.log 2>&1; fi
PMCGRP=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
if [ -n "${PMC_TRAFFIC:-}" ]; then PMCGRP=("FETCH_SIZE" "WRITE_SIZE"); fi
if [ -n "${PMC_ONLY:-}" ]; then PMCGRP=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"); fi
for grp in "${PMCGRP[@]}"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-include-regex "${KREGEX:-bt_search}" --output-format csv -d $OUT/pmc_$name -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu > $OUT/pmc_$name.log 2>&1
done
# summarise
python - "$OUT" <<'PY'
import sys, os, csv, glob, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats:", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print("  ", row)
for d in sorted(glob.glob(out + "/pmc_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if os.environ.get("KREGEX", "bt_search") in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("  PMC %-32s per-dispatch %s" % (k, ["%.4g" % x for x in v]))
PY
