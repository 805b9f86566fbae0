#!/bin/bash
# Twelfth GPU call of round 5: bowtie-amd's batch size.  With 4 M-read batches a launch lasts 0.34 s, carry-over lets a read
# ride along for 12 launches = 4 s, and the heaviest reads need more: every launch then waits for the stragglers of the batch
# twelve launches back (eleventh call: one batch per 1.05 s = 3.8 M reads/s, GPU-side).  Larger batches give the stragglers
# their time.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_12; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
BT_CLI_TIMELINE=0 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref --extra "--batch 8388608" > $O/cli_64m_b8.json 2> $O/cli_64m_b8.err
python - "$O/cli_64m_b8.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads, --batch 8 M, as the binary decides (row space; 4 M batches: 18.33 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
for b in 8388608 16777216; do
	t0=$(date +%s.%N)
	BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch $b -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_b$b.err
	t1=$(date +%s.%N)
	python - "$t0" "$t1" "$b" "$O/cli_192m_b$b.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
print("bowtie-amd 192 M reads (the 64 M-read file three times; SAM to /dev/null), --batch %s (4 M batches: 49.19 s = 3.90 M reads/s): %.2f s = %.2f M reads/s" % (sys.argv[3], t, 192.0 / t))
err = open(sys.argv[4], errors="replace").read().splitlines()
print("\n".join("   " + l for l in err if "Stage busy" in l or "Time" in l or "at least one" in l))
tl = [l for l in err if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
	grep "\[timeline\]" $O/cli_192m_b$b.err > $O/cli_192m_b${b}_timeline.txt
done
cat $S
