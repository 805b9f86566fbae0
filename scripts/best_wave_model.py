#!/usr/bin/env python3
"""How well are a wavefront's 64 lanes used by bt_best_kernel's loop in the BF_REFILL build -- lanes take a new read when
`least` of them wait for one -- against the shipped loop, which hands out reads a wavefront at a time (least = 64)?
No GPU: a patched copy of the emulator (under /tmp) goes through the loop with 64 lanes side by side (tests/emu/bt_emu.cpp:
emu_best_wave) and counts turns of the loop, turns in which some lane began a read (the wavefront pays for the tree and
every leaf's set-up in those), and lane steps; every turn counts the same, which it does not on the GPU.

  python scripts/best_wave_model.py [--workload n2_best|pe] [--reads 3000]
"""
import argparse
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import best_arena_reads as BAR                      # noqa: E402  (the child program that runs a workload through a library)
W = "/tmp/bt_best_wave_model"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="n2_best")
    ap.add_argument("--reads", type=int, default=3000)
    a = ap.parse_args()
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W + "/tests")
    shutil.copytree(os.path.join(ROOT, "tests", "emu"), W + "/tests/emu", ignore=shutil.ignore_patterns("*.so"))
    shutil.copytree(os.path.join(ROOT, "bowtie_amd", "csrc"), W + "/bowtie_amd/csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), W + "/include")
    p = W + "/tests/emu/bt_emu.cpp"
    s = open(p).read()
    for old, new in [
        ("\tuint32_t next = 0;\n\tfor (;;) {\n\t\tuint32_t working = 0, waiting = 0, alive = 0;",
         "\tuint32_t next = 0; unsigned long long turns = 0, steps = 0, beginTurns = 0;\n\tif (getenv(\"WV_LEAST\")) least = (uint32_t)atoi(getenv(\"WV_LEAST\"));\n"
         "\tfor (;;) {\n\t\tturns++; bool anyBegin = false;\n\t\tuint32_t working = 0, waiting = 0, alive = 0;"),
        ("\t\t\tif (w >= n) drained[l] = 1; else bf_run_begin(XS[l], B, w, R[l], kind);",
         "\t\t\tif (w >= n) drained[l] = 1; else { bf_run_begin(XS[l], B, w, R[l], kind); anyBegin = true; }"),
        ("\t\t\tif (!bf_run_step(XS[l], B, R[l])) bf_run_end(XS[l], B, R[l]);\n\t\t}\n\t}",
         "\t\t\tsteps++; if (!bf_run_step(XS[l], B, R[l])) bf_run_end(XS[l], B, R[l]);\n\t\t}\n\t\tif (anyBegin) beginTurns++;\n\t}\n"
         "\tfprintf(stderr, \"[wave] lanes %zu, a read is taken when %u wait: %llu turns (%llu with a begin), %llu lane steps, lanes used %.3f\\n\", W, least, turns, beginTurns, steps, (double)steps / ((double)turns * W));"),
        ("in->n_reads < 24u ? (in->n_reads ? in->n_reads : 1u) : 24u", "in->n_reads < 64u ? (in->n_reads ? in->n_reads : 1u) : 64u"),
        ("in1->n_reads < 24u ? (in1->n_reads ? in1->n_reads : 1u) : 24u", "in1->n_reads < 64u ? (in1->n_reads ? in1->n_reads : 1u) : 64u"),
    ]:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    open(p, "w").write(s)
    lib = W + "/libbt_emu_wave.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-DBF_FAST_EXTEND=1", "-o", lib, p, W + "/bowtie_amd/csrc/bt_host.cpp"])
    for least in (1, 8, 16, 32, 64):
        r = subprocess.run([sys.executable, "-c", BAR.CHILD % dict(root=ROOT, lib=lib, workload=a.workload, reads=a.reads)],
                           env=dict(os.environ, WV_LEAST=str(least)), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if least == 1:
            sys.stdout.write(r.stdout.decode())
        for l in r.stderr.decode().splitlines():
            if l.startswith("[wave]"):
                print(l[7:] + ("   <- the shipped loop" if least == 64 else ""))


if __name__ == "__main__":
    main()
