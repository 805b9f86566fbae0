#!/bin/bash
# First GPU call of round 4.
#   gpurun --timeout 1500 -- 'bash scripts/r4/first.sh'
# 1. The rare wrong mismatch list of the streamed path: scripts/r4/stream_stress.py, six processes side by
#    side, in four settings -- as shipped; staging pool poisoned; poisoned + what the device holds afterwards compared with what
#    was delivered; every copy-back ordered behind the search stream.
# 2. bt_best_kernel with -DBF_FAST_EXTEND=1 (bit-identical in the host build, never run on a GPU): the best-first and paired
#    GPU tests through it, then the best-first workloads against the default library.
# 3. The binary end to end on the hg19-scale index with BT_CLI_TIMELINE=1, then with BT_CLI_PINNED=1; outputs compared.
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('%.3f M reads/s, %.1f ms/step' % (d['value']/1e6, d['ms_per_step']))" 2>&1 | tail -1; }

# ---- 1. streamed path under load ----
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
	grep -h "stream-recheck" $O/stress_$name.*.err 2>/dev/null | head -5 >> $S
}
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "streamed or carry" > $O/streamed_tests.txt 2>&1
say "streamed / carry-over tests with the staging area's own pool cursor: $(tail -1 $O/streamed_tests.txt)"
# the load the failure was seen under: other tests' kernels (whole-GPU grids) from two more processes, for all of part 1
LOADPG=""
for w in 1 2; do setsid bash -c 'while true; do timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 0 -k "best_first_vs_oracle_ragged" > /dev/null 2>&1; done' & LOADPG="$LOADPG $!"; done
stress shipped 70
stress poison 70 BT_STREAM_POISON=1
stress poison_recheck 60 BT_STREAM_POISON=1 BT_STREAM_RECHECK=1
stress ordered 60 BT_STREAM_POISON=1 BT_STREAM_ORDERED=1
stress mmsort 45 BT_STREAM_POISON=1 BT_LIB=libbowtie_amd_mmsort.so
for pg in $LOADPG; do kill -- -$pg 2>/dev/null; done
sleep 2

# ---- 2. fast extend ----
BT_LIB=libbowtie_amd_fastext.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "best or paired or config5 or v3 or M3 or strata" > $O/fastext_parity.txt 2>&1
say "fast-extend library, best-first / paired GPU tests: $(tail -1 $O/fastext_parity.txt)"
for lib in libbowtie_amd.so libbowtie_amd_fastext.so libbowtie_amd_fastext_ng.so libbowtie_amd_fastext_nr.so; do
	for wl in ecoli_n2_best_100 ecoli_pe_n1_best_50; do f=$O/bench_${wl}_$lib; BT_LIB=$lib timeout 120 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib $wl (round 3: 6.6 / 30.8 M): $(val $f.json)"; done
done
for lib in libbowtie_amd.so libbowtie_amd_fastext.so; do
	f=$O/bench_big_pe_$lib; BT_LIB=$lib timeout 300 python bench.py --workload big_pe_n1_best_50 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_pe_n1_best_50 (round 3: 3.29 M reads/s): $(val $f.json)"
	f=$O/bench_big_n2_best_$lib; BT_LIB=$lib timeout 300 python bench.py --workload big_n2_best_100 --reads 16000000 --steps 2 --warmup 1 --no-cpu --also none > $f.json 2> $f.log; say "$lib big_n2_best_100 16 M reads (round 3: 0.85 M reads/s): $(val $f.json)"
done

# ---- 3. the binary, file to file ----
BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 32000000 --no-ref > $O/cli_timeline.json 2> $O/cli_timeline.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_timeline.json").read().strip().splitlines()[-1])
print("bowtie-amd 32 M reads, default: %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join(l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
PY
cp /tmp/cli_ours.sam /tmp/cli_default.sam 2>/dev/null
BT_CLI_PINNED=1 BT_CLI_TIMELINE=1 timeout 300 python scripts/cli_bench.py --index big --reads 32000000 --no-ref > $O/cli_pinned.json 2> $O/cli_pinned.err
python - >> $S <<PY
import json
d = json.loads(open("$O/cli_pinned.json").read().strip().splitlines()[-1])
print("bowtie-amd 32 M reads, BT_CLI_PINNED=1: %.2f s = %.2f M reads/s" % (d["bowtie_amd_s"], d["bowtie_amd_reads_per_s"] / 1e6))
print("\n".join(l for l in d["bowtie_amd_stderr"] if "Stage busy" in l or "Time" in l))
PY
if cmp -s <(grep -v '^@PG' /tmp/cli_default.sam) <(grep -v '^@PG' /tmp/cli_ours.sam); then say "pinned run's SAM = default run's SAM"; else say "pinned run's SAM DIFFERS from the default run's"; fi
timeout 300 python scripts/cli_bench.py --index big --reads 32000000 --no-ref --extra=--no-stream > $O/cli_nostream.json 2> $O/cli_nostream.err
if cmp -s <(grep -v '^@PG' /tmp/cli_default.sam) <(grep -v '^@PG' /tmp/cli_ours.sam); then say "--no-stream run's SAM = streamed run's SAM (32 M reads)"; else say "--no-stream run's SAM DIFFERS from the streamed run's"; fi
cat $S
