#!/bin/bash
# Third GPU call of round 4 (the second never ran: driver-side error before a box was had) = second.sh + item 2b.
#   gpurun --timeout 1700 -- 'bash scripts/r4/third.sh'
# 2b. bt_search_kernel variants (csrc/Makefile searchvariants): parity subset, then big_n2_100 at 16 M reads per step,
#    carry-over 12: mmregs / skipdead / mmskip / defer at three gate settings.
# 1. The streamed stress with the mismatch-pool cursors of rounds 2-3 (BT_STREAM_OLD_CURSOR=1): does the rare failure
#    come back when the round-4 change (a staging area's own cursor) is taken out?
# 2. BF_FAST_EXTEND without BF_REFILL (the variant that was 1.9x / 1.7x on e_coli in the first call): parity, big workloads.
# 3. big_n2_100 at 200 M reads per step with carry-over (never measured with the second design).
# 4. bowtie-amd, 64 M reads, carry-over ages 2 / 3 / 4 / 12 (the first call's timeline: with 12 no batch comes back before the end).
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('%.3f M reads/s (%.3f M aligned), %.1f ms/step, frac %.4f' % (d.get('reads_processed_per_s', d['value'])/1e6, d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1; }

stress() {   # name, seconds, env...
	local name=$1 secs=$2; shift 2
	local pids=""
	for w in 1 2 3 4 5 6; do ( env "$@" timeout $((secs + 120)) python scripts/r4/stream_stress.py --seconds $secs --tag $name.$w > $O/stress_$name.$w.json 2> $O/stress_$name.$w.err ) & pids="$pids $!"; done
	wait $pids
	python - "$name" >> $S <<PY
import json, glob, sys
name = sys.argv[1]
rounds = fails = 0
first = None
for f in sorted(glob.glob("$O/stress_%s.*.json" % name)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print("  %s: no result (%s)" % (f, e)); continue
    rounds += d["rounds"]; fails += d["fails"]
    if d["reports"] and first is None:
        first = d["reports"][0]
print("stress %-22s %5d rounds, %d failed" % (name, rounds, fails))
if first:
    for w in first["what"]:
        if isinstance(w, dict):
            print("   first failure: round %d carry %d batch %d (%s): %d reads, kinds %s" % (first["round"], first["carry"], w["batch"], w["reads"], w["n_bad"], w["kinds"]))
            for x in w["first"][:2]:
                print("     read %d mm %s pool %s\n       got  %s\n       want %s" % (x["i"], x["mm"], x["pool"], x["got"][:200], x["want"][:200]))
        else:
            print("   first failure:", w)
PY
}
LOADPG=""
for w in 1 2; do setsid bash -c 'while true; do timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "best_first_vs_oracle_ragged" > /dev/null 2>&1; done' & LOADPG="$LOADPG $!"; done
stress oldcursor 75 BT_STREAM_OLD_CURSOR=1
stress oldcursor_noperm 45 BT_STREAM_OLD_CURSOR=1 STRESS_NO_PERMUTE=1
for pg in $LOADPG; do kill -- -$pg 2>/dev/null; done
sleep 2

BT_LIB=libbowtie_amd_fastext_nr.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "best or paired or config5 or v3 or M3 or strata" > $O/fastext_nr_parity.txt 2>&1
say "fast-extend (no refill) library, best-first / paired GPU tests: $(tail -1 $O/fastext_nr_parity.txt)"
for lib in libbowtie_amd_fastext_nr.so; do
	f=$O/bench_big_pe_$lib; BT_LIB=$lib timeout 300 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_pe_n1_best_50 (default library, first call: 3.20 M reads/s): $(val $f.json)"
	f=$O/bench_big_n2_best_$lib; BT_LIB=$lib timeout 300 python bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_n2_best_100 16 M reads (default library, first call: 0.847 M reads/s): $(val $f.json)"
done


# ---- 2b: search-kernel variants
for lib in libbowtie_amd_mmskip.so libbowtie_amd_defer.so; do
	BT_LIB=$lib BT_SLOW_PERIOD=3 BT_SLOW_MIN=12 timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not best and not paired" > $O/parity_$lib.txt 2>&1
	say "$lib (gate 3/12 where it has one) unpaired non-best GPU parity tests: $(tail -1 $O/parity_$lib.txt)"
done
ab() {   # tag lib env...
	local tag=$1 lib=$2; shift 2
	local f=$O/ab_$tag
	env BT_LIB=$lib "$@" timeout 300 python bench.py --reads 16000000 --carry 12 --steps 6 --warmup 2 --no-cpu --also none > $f.json 2> $f.log
	say "A/B $tag big_n2_100 16 M reads/step carry 12: $(val $f.json)"
}
ab default libbowtie_amd.so
ab mmregs libbowtie_amd_mmregs.so
ab skipdead libbowtie_amd_skipdead.so
ab mmskip libbowtie_amd_mmskip.so
ab defer_open libbowtie_amd_defer.so
ab defer_p2_m16 libbowtie_amd_defer.so BT_SLOW_PERIOD=2 BT_SLOW_MIN=16
ab defer_p3_m12 libbowtie_amd_defer.so BT_SLOW_PERIOD=3 BT_SLOW_MIN=12
ab defer_p4_m24 libbowtie_amd_defer.so BT_SLOW_PERIOD=4 BT_SLOW_MIN=24
ab default_again libbowtie_amd.so

f=$O/bench_big_n2_100_200M_carry2; timeout 400 python bench.py --carry 2 --steps 4 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "big_n2_100, 200 M reads per step, carry-over 2, 4 steps (round 3 without: 12.35-12.54 M reads/s): $(val $f.json)"
python - >> $S <<PY
import json
d = json.loads(open("$O/bench_big_n2_100_200M_carry2.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("   kernel_ms_avg %.1f flush_ms_total %s verified %s" % (r["kernel_ms_avg"], d["config"].get("flush_ms_total"), d["config"].get("hits_verified_against_text")))
PY

for age in 3 2 4 12; do
	BT_CLI_CARRY=$age BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 64000000 --no-ref > $O/cli_carry$age.json 2> $O/cli_carry$age.err
	python - >> $S <<PY
import json
d = json.loads(open("$O/cli_carry$age.json").read().strip().splitlines()[-1])
print("bowtie-amd 64 M reads, carry-over $age: %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join("   " + l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
tl = [l for l in d["bowtie_amd_stderr"] if "results back" in l]
print("   first results back: %s; last: %s" % (tl[0].split()[1] if tl else "?", tl[-1].split()[1] if tl else "?"))
PY
	md5sum /tmp/cli_ours.sam | cut -c1-32 >> $S
done
cat $S
