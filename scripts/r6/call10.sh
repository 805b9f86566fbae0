#!/bin/bash
# GPU call 10 of round 6: the ring of batches 16 -> 64 (bt_core.h) -- reads may ride along for up to 62 launches, bowtie-amd lets them
# ride 22 with 24 batches in flight on this host -- and the writer's large pieces past stdio's buffer.  Carry-over / stream / binary
# tests, the default bench for two steps (the kernel's instructions are the ones measured before, bar the width of a bit field:
# profiles/r6/ring64_isa_diff.txt), then the binary on 192 M reads, each run a few seconds after the one before (the driver is still
# releasing the last process's HBM when the next one starts right behind it: call 9).
#   gpurun --timeout 1800 -- 'bash scripts/r6/call10.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_10; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zzz_gpu_stress.py tests/test_gpu_cli.py -q -m gpu -k "carry or stream or cli or ticks" > $O/gpu_tests.txt 2>&1
say "pytest -m gpu -k 'carry or stream or cli or ticks' (parity, stress, binary), $(( $(date +%s) - t0 )) s: $(tail -1 $O/gpu_tests.txt)"
grep -h "^FAILED" $O/gpu_tests.txt | head -5 | tee -a $S
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --also none > $O/bench.json 2> $O/bench.log
python - "$O/bench.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("python bench.py --steps 2 --warmup 1: %.3f M reads processed/s, kernel %s avg %.1f ms, frac %.4f, traffic %s, verified %s" % (d["reads_processed_per_s"] / 1e6, r["kernel"], r["kernel_ms_avg"], r["frac"], r.get("traffic"), d["config"].get("hits_verified_against_text")))
except Exception as e:
    print("bench: FAILED (%s)" % e)
PY
BT_CLI_TIMELINE=0 timeout 900 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - "$O/cli_64m.json" >> $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bowtie-amd 64 M reads file -> SAM file (round 5: 16.04 s; final call: 14.51 s; call 9: 12.79 s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
except Exception as e:
    print("cli 64 M: FAILED (%s)" % e)
PY
FQ=/tmp/cli_bench_big_64000000.fq
BASE=$(ls /tmp/bowtie_amd_idx/*.1.ebwt | grep -v rev | head -1 | sed 's/.1.ebwt//')
if [ -f $FQ ]; then
	run() {   # label, file tag, seconds to wait first, extra arguments ("-" = none), environment...
		local label="$1" tag="$2" wait="$3" extra="$4"; shift 4
		[ "$extra" = "-" ] && extra=""
		sleep $wait
		env "$@" timeout 400 python scripts/r6/cli_run.py "$label" $O/cli_192m_$tag.err 192 -- bowtie_amd/bowtie-amd -p 64 -t -S -n 2 $extra -x $BASE $FQ,$FQ,$FQ /dev/null >> $S
		tail -1 $S
	}
	run "192 M reads -> /dev/null, the tree's defaults" default 10 - A=1
	run "... ride 12, 13 in flight (round 5's)" c12f13 10 - BT_CLI_CARRY=12 BT_CLI_INFLIGHT=13
	run "... ride 12, 14 in flight" c12f14 10 - BT_CLI_CARRY=12 BT_CLI_INFLIGHT=14
	run "... ride 30, 32 in flight" c30f32 10 - BT_CLI_CARRY=30 BT_CLI_INFLIGHT=32
	run "... batches of 8 M reads" b8m 10 "--batch 8388608" A=1
	run "... batches of 16 M reads" b16m 10 "--batch 16777216" A=1
	run "... defaults, started right behind the last" default_b2b 0 - A=1
	run "... defaults again" default2 10 - A=1
	head -n 16000000 $FQ > /tmp/r4m.fq
	for cfg in "A=1" "BT_CLI_CARRY=12 BT_CLI_INFLIGHT=13" "BT_CLI_CARRY=40"; do
		env $cfg bowtie_amd/bowtie-amd -p 64 -S -n 2 --batch 100000 -x $BASE /tmp/r4m.fq /tmp/r4m.sam 2> $O/r4m.err
		say "4 M reads in batches of 100 000 -> SAM file, $cfg: md5 without @PG $(grep -v '^@PG' /tmp/r4m.sam | md5sum | cut -c1-32), $(grep -vc '^@' /tmp/r4m.sam) records"
	done
	rm -f /tmp/r4m.sam /tmp/r4m.fq
	grep -a "timeline" $O/cli_192m_default.err > $O/cli_192m_default_timeline.txt
fi
cat $S
