#!/bin/bash
# Sixteenth GPU call of round 5 (call 15 killed its own FASTQ generator with a short timeout and measured nothing): bowtie-amd on
# 192 M reads with 16 M reads per batch and a carry-over age of 4 -- four launches of 1.3 s are the 5 s the heaviest reads need.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_16; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
python scripts/cli_bench.py --index big --reads 64000000 --no-ref --extra "--batch 16777216" > $O/cli_64m_b16.json 2> $O/cli_64m_b16.err
python - "$O/cli_64m_b16.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads, --batch 16 M, carry-over 12 (8 M batches: 16.04 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
ls -la $FQ | tee -a $S
t0=$(date +%s.%N)
BT_CLI_CARRY=4 BT_CLI_TIMELINE=1 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 --batch 16777216 -x $BASE $FQ,$FQ,$FQ /dev/null 2> $O/cli_192m_b16_c4.err
t1=$(date +%s.%N)
python - "$t0" "$t1" "$O/cli_192m_b16_c4.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
err = open(sys.argv[3], errors="replace").read().splitlines()
tl = [l for l in err if "results back" in l]
print("bowtie-amd 192 M reads, --batch 16 M, carry-over age 4 (age 12: 34.4 s): %.2f s = %.2f M reads/s; first results back %s, last %s; %s" % (
    t, 192.0 / t, tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?", "; ".join(l.strip() for l in err if "Stage busy" in l or "at least one" in l)))
PY
cat $S
