#!/bin/bash
# Eleventh GPU call of round 5: (1) the default workload with carry-over between its steps (round 4: worse at 200 M reads per
# step; the kernel has changed); (2) bowtie-amd end to end: 64 M reads as the binary now decides (no locus image for an input
# that small), and 192 M reads (the 64 M file three times on the command line: the image pays from ~100 M reads on).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_11; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, kernel %s avg %.1f ms (flush total %.0f ms), frac %.4f, verified %s %s; carry %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r.get('flush_ms_total', 0), r['frac'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit'), r.get('carry_over_launches')))" 2>&1 | tail -1; }
f=$O/carry_200m; timeout 400 python bench.py --carry 12 --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log
say "big_n2_100 200 M reads per step, carry-over 12, 3 steps (without: 15.35 M reads/s): $(val $f.json)"
BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads, as the binary decides (call 10 with the image: 18.79 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l or l.rstrip().endswith(" end")))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
df -h /tmp | tail -1 | tee -a $S
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(python -c "print('/tmp/bowtie_amd_idx/synth_2860000000')")
ls $BASE.1.ebwt > /dev/null 2>&1 || BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
OUT=/tmp/cli_192m.sam
avail=$(df --output=avail -BG /tmp | tail -1 | tr -dc '0-9')
if [ "${avail:-0}" -lt 60 ]; then OUT=/dev/null; fi
t0=$(date +%s.%N)
BT_CLI_TIMELINE=1 timeout 400 bowtie_amd/bowtie-amd -p 64 -t -S -n 2 -x $BASE $FQ,$FQ,$FQ $OUT 2> $O/cli_192m.err
t1=$(date +%s.%N)
python - "$t0" "$t1" "$OUT" "$O/cli_192m.err" >> $S <<'PY'
import sys
t = float(sys.argv[2]) - float(sys.argv[1])
print("bowtie-amd 192 M reads (the 64 M-read file three times; SAM to %s): %.2f s = %.2f M reads/s" % (sys.argv[3], t, 192e6 / t))
err = open(sys.argv[4], errors="replace").read().splitlines()
print("\n".join("   " + l for l in err if "Stage busy" in l or "Time" in l or "reads processed" in l or "at least one" in l or l.rstrip().endswith(" end")))
tl = [l for l in err if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
grep "\[timeline\]" $O/cli_192m.err > $O/cli_192m_timeline.txt
cat $S
