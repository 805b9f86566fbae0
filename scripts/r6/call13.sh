#!/bin/bash
# GPU call 13 of round 6: (a) does the SAM formatter scale with threads on this host the way bowtie-amd's writer uses it
# (scripts/r6/fmt_probe.cpp; the writer takes 0.19 s per batch of 12 M reads on 128 threads that is 0.03 s of work each)?
# (b) the new bt_align_stream_room and the binary's tests.
#   gpurun --timeout 900 -- 'bash scripts/r6/call13.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_13; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
say "host: $(nproc) hardware threads; $(lscpu | grep -E 'Model name|Socket|NUMA node\(s\)' | tr -s ' ' | tr '\n' ';')"
g++ -O3 -std=c++17 -fPIC -c bowtie_amd/csrc/bt_io.cpp -o /tmp/bt_io.o 2> $O/probe_build.err && g++ -O3 -std=c++17 -Ibowtie_amd/csrc scripts/r6/fmt_probe.cpp /tmp/bt_io.o -lz -lpthread -o /tmp/fmt_probe 2>> $O/probe_build.err
timeout 300 /tmp/fmt_probe 12582912 1 2>&1 | sed 's/^/   /' | tee -a $S
timeout 300 /tmp/fmt_probe 12582912 0 2>&1 | sed 's/^/   /' | tee -a $S
command -v numactl > /dev/null && { say "   interleaved over the NUMA nodes:"; timeout 300 numactl --interleave=all /tmp/fmt_probe 12582912 1 2>&1 | sed 's/^/   /' | tee -a $S; }
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py -q -m gpu -k "cli or stream or carry" > $O/gpu_tests.txt 2>&1
say "pytest -m gpu -k 'cli or stream or carry': $(tail -1 $O/gpu_tests.txt)"
grep -h "^FAILED" $O/gpu_tests.txt | head -5 | tee -a $S
cat $S
