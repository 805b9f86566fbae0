#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the probe kernel (known bytes) -> calibration for the search kernel's traffic.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-include-regex "bt_probe_rank" --output-format csv -d $O/$name -- python $R/scripts/calib_fetch.py > $O/$name.log 2>&1
done
python - "$O" <<'PY'
import sys, csv, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "bt_probe_rank" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s per dispatch %s" % (k, ["%.6g" % x for x in v]))
PY
grep -h "probes per launch" $O/*.log | head -1
