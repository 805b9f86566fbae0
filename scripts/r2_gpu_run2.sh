#!/bin/bash
# round-2 GPU pass 2: new GPU tests (index family, gated device path, on-stream retry, carry-over) and the carry-over A/B
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_cli.py -m gpu -x -q -k "carry or device_path or streamed or scale or 100mbp or cli" ) > $O/gputests_new.txt 2>&1
tail -15 $O/gputests_new.txt
grep -q " passed" $O/gputests_new.txt && ! grep -q "failed\|Aborted\|error" $O/gputests_new.txt || { echo "tests failed: stopping"; grep -B5 -A60 "Error\|FAILED\|assert" $O/gputests_new.txt | head -200; exit 1; }
run() { # name, args...
  local name=$1; shift
  timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$name: %.2f M reads/s  ms/step %.0f  kernel_ms_avg %.0f main %.0f flush %.0f  active %.1f  %s" % (j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r.get("kernel_ms_main_avg", 0), r.get("flush_ms_total", 0), r["mean_active_lanes_per_round"], r["kernel"]))
except Exception as e:
    print("$name failed", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run b16_carry_p1    --reads 16000000 --steps 6 --warmup 2 --pipes 1 --no-cpu --no-verify
run b16_nocarry_p1  --reads 16000000 --steps 6 --warmup 2 --pipes 1 --no-cpu --no-verify --no-carry
run b16_carry_p2    --reads 16000000 --steps 6 --warmup 2 --pipes 2 --no-cpu --no-verify
run b4_carry_p1     --reads 4000000 --steps 12 --warmup 2 --pipes 1 --no-cpu --no-verify
run b200_carry      --no-cpu
run b200_nocarry    --no-cpu --no-verify --no-carry
