#!/bin/bash
# round-2 GPU pass 4: multi-launch carry-over -- tests, then 16 M-read launches with it, then the CLI end to end
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 -k "carry or device_path or streamed" ) > $O/gputests_new.txt 2>&1
tail -6 $O/gputests_new.txt
grep -q " passed" $O/gputests_new.txt && ! grep -q "failed\|Aborted\|error\|Timeout" $O/gputests_new.txt || { echo "tests failed: stopping"; grep -B5 -A60 "Error\|FAILED\|assert\|Timeout" $O/gputests_new.txt | head -150; exit 1; }
( time timeout 300 python -m pytest tests/test_gpu_cli.py -m gpu -x -q --timeout 120 ) > $O/gputests_cli.txt 2>&1
tail -4 $O/gputests_cli.txt
run() { # name, args...
  local name=$1; shift
  timeout 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$name: %.2f M reads/s  ms/step %.0f  kernel_ms_avg %.0f main %.0f flush %.0f  active %.1f  %s" % (j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r.get("kernel_ms_main_avg", 0), r.get("flush_ms_total", 0), r["mean_active_lanes_per_round"], r["kernel"]))
except Exception as e:
    print("$name failed", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run b16_carry12_p1    --reads 16000000 --steps 10 --warmup 2 --pipes 1 --no-cpu
timeout 300 python scripts/cli_bench.py --index big --reads 24000000 --no-ref --extra "--batch 4194304 -t" > $O/cli_big_24m.json 2> $O/cli_big_24m.err; tail -c 700 $O/cli_big_24m.json; tail -3 $O/cli_big_24m.err
