#!/bin/bash
# Fifth GPU call of round 4: the wavefront automaton of bt_best_kernel (default) against the call-by-call kernel
# (BT_BEST_NESTED=1), its gates swept in one process per workload (bench.py --env-sweep); parity first.  Then bowtie-amd
# with the searcher that waits instead of flushing.
#   gpurun --timeout 900 -- 'bash scripts/r4/fifth.sh'
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "best or paired or config5 or v3 or M3 or strata" > $O/parity_automaton.txt 2>&1
say "automaton (default) best-first / paired GPU tests: $(tail -1 $O/parity_automaton.txt)"
BT_BEST_NESTED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "best_first or paired" > $O/parity_nested.txt 2>&1
say "call-by-call kernel (BT_BEST_NESTED=1) best-first / paired GPU tests: $(tail -1 $O/parity_nested.txt)"
SW_E="nested:BT_BEST_NESTED=1;c8:BT_BEST_COLD_MIN=8;c32:BT_BEST_COLD_MIN=32;c48:BT_BEST_COLD_MIN=48;s2:BT_BEST_SEND_PERIOD=2,BT_BEST_SEND_MIN=16;s4:BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;t32:BT_BEST_TAKE_MIN=32;t4:BT_BEST_TAKE_MIN=4;c32s4:BT_BEST_COLD_MIN=32,BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;c32t32:BT_BEST_COLD_MIN=32,BT_BEST_TAKE_MIN=32"
SW_B="nested:BT_BEST_NESTED=1;c32:BT_BEST_COLD_MIN=32;s4:BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;c32s4:BT_BEST_COLD_MIN=32,BT_BEST_SEND_PERIOD=4,BT_BEST_SEND_MIN=24;c8:BT_BEST_COLD_MIN=8"
run() {   # workload sweep extra...
	local wl=$1 sw=$2; shift 2
	local f=$O/ab_$wl
	timeout 420 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --also none --env-sweep "$sw" "$@" > $f.json 2> $f.log
	say "== $wl $*"
	python -c "import json; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); print('   default (automaton): %.3f M reads/s, kernel %.1f ms, verified %s %s' % (d.get('reads_processed_per_s', d['value'])/1e6, d['roofline']['kernel_ms_avg'], d['config'].get('hits_verified_against_text'), d['config'].get('verified_unit')))" 2>&1 | tail -1 | tee -a $S
	grep "env-sweep" $f.log | sed 's/^\[bench\] /   /' | tee -a $S
}
run ecoli_n2_best_100 "$SW_E"
run ecoli_pe_n1_best_50 "$SW_E"
run big_pe_n1_best_50 "$SW_B"
run big_n2_best_100 "$SW_B" --reads 8000000
BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_64m.json 2> $O/cli_64m.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_64m.json").read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads (third call: 18.56 s = 3.45 M reads/s): %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l or "teardown" in l or " end" in l))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
cat $S
