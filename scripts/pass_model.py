#!/usr/bin/env python3
"""What a lock-step round of bt_search_kernel costs a wavefront is the number of times its lanes go round bt_lane_run's loop
and through the slow-state sweep in that round: every trip is paid for by all 64 lanes, whoever needed it.  This counts them
on the CPU: the host build of the automaton (tests/emu; a patched copy under /tmp) records per lane and round the trips
round the loop, the sweeps and the locus blocks; a wavefront's round costs the maximum over its lanes.

  python scripts/pass_model.py [--reads 20000] [--synthetic 30000000] [--mode n2]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
W = "/tmp/bt_pass_model"


def build_probe():
    os.makedirs(W + "/emu", exist_ok=True)
    s = open(os.path.join(ROOT, "bowtie_amd/csrc/bt_core.h")).read()
    old = "#define BT_PROF_T0(v)\n#define BT_PROF_ADD(k, v)\n#define BT_PROF_PASS()\n#define BT_PROF_TICK(k)\n#endif"
    assert old in s
    s = s.replace(old, "#define BT_PROF_T0(v)\n#define BT_PROF_ADD(k, v)\nextern \"C\" { extern unsigned bt_emu_ticks[4]; }\n#define BT_PROF_PASS() (bt_emu_ticks[0]++)\n"
                       "#define BT_PROF_TICK(k) (bt_emu_ticks[(k) == PS_RUN_ITERS ? 1 : (k) == PS_LOCUS_PASSES ? 2 : 3]++)\n#endif")
    old = "\twhile (BT_IS_SLOW(L.state) && req.kind == RQ_NONE) {\n\t\tBT_PROF_PASS();"
    assert old in s
    s = s.replace(old, "\tunsigned sweep_no_ = 0;\n\twhile (BT_IS_SLOW(L.state) && req.kind == RQ_NONE) {\n\t\tBT_PROF_PASS();\n\t\tif (sweep_no_++ > 0) bt_emu_again[L.state < 64 ? L.state : 63]++;")
    s = s.replace('extern "C" { extern unsigned bt_emu_ticks[4]; }', 'extern "C" { extern unsigned bt_emu_ticks[4]; extern unsigned long long bt_emu_again[64]; }')
    open(W + "/bt_core.h", "w").write(s)
    e = open(os.path.join(ROOT, "tests/emu/bt_emu.cpp")).read()
    e = e.replace('#include "../../bowtie_amd/csrc/bt_host.h"', '#include "../bt_host.h"')
    e = e.replace("/* Same contract as bt_align_batch (host pointers); nLanes lock-step lanes. */",
                  'extern "C" { unsigned bt_emu_ticks[4] = {0, 0, 0, 0}; unsigned long long bt_emu_again[64]; }\nextern "C" unsigned long long* emu_again() { return bt_emu_again; }\nstatic std::vector<unsigned> g_tr;\n'
                  'extern "C" unsigned* emu_ticks(size_t* n) { *n = g_tr.size(); return g_tr.data(); }\n')
    old = "\twhile (live > 0) {\n\t\tfor (uint32_t g = 0; g < nLanes; g++) {\n\t\t\tif (drained[g]) continue;"
    assert old in e
    e = e.replace(old, "\tg_tr.clear();\n\twhile (live > 0) {\n\t\tconst size_t row = g_tr.size();\n\t\tg_tr.resize(row + (size_t)nLanes * 4, 0u);\n"
                       "\t\tfor (uint32_t g = 0; g < nLanes; g++) {\n\t\t\tbt_emu_ticks[0] = bt_emu_ticks[1] = bt_emu_ticks[2] = bt_emu_ticks[3] = 0;\n\t\t\tif (drained[g]) continue;")
    old = "\t\t\tCNT[CN_TLFEX] += req.tally >> 16;"
    assert old in e
    e = e.replace(old, "\t\t\tfor (int k = 0; k < 4; k++) g_tr[row + (size_t)g * 4 + k] = bt_emu_ticks[k] | (k == 3 ? 0x80000000u : 0u);\n" + old, 1)
    open(W + "/emu/bt_emu.cpp", "w").write(e)
    for f in ("bt_rank.h", "bt_best.h", "bt_host.h", "bt_host.cpp", "bt_kernels.h", "bt_io.h"):
        t = open(os.path.join(ROOT, "bowtie_amd/csrc", f)).read().replace("../../include/bowtie_amd.h", os.path.join(ROOT, "include/bowtie_amd.h"))
        open(os.path.join(W, f), "w").write(t)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", W + "/libbt_emu_pm.so", W + "/emu/bt_emu.cpp", W + "/bt_host.cpp"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--synthetic", type=int, default=30000000)
    ap.add_argument("--mode", default="n2")
    ap.add_argument("--lanes", type=int, default=256)
    a = ap.parse_args()
    build_probe()
    import emu_lib as E
    E.LIB_PATH = W + "/libbt_emu_pm.so"
    E.SRCS = []
    from bowtie_amd import _abi as A
    from bowtie_amd.synth import synth_reads
    import common as T
    if a.synthetic:
        import torch
        from bowtie_amd import ebwt_build as EB
        index, text, note = EB.ensure_big_index(a.synthetic, torch.device("cpu"), cache_dir="/tmp/bt_textmode_model")
    else:
        index = os.path.join(T.G, "e_coli"); text = T.joined_text("e_coli")
    batch = synth_reads(text, a.reads, a.len, mm_dist=(0, 1, 2, 2, 3, 4), seed=11)
    L = E.lib()
    L.emu_ticks.restype = C.POINTER(C.c_uint32)
    L.emu_ticks.argtypes = [C.POINTER(C.c_size_t)]
    emu = E.EmuAligner(index)
    for label, off in (("locus mode", "0"), ("row space", "1")):
        os.environ["EMU_LOCUS_OFF"] = off
        cnt = A.OpCounts()
        emu.align(A.make_policy(**T.MODES[a.mode]), batch, n_lanes=a.lanes, lite=True, pal_cap=16384, counts=cnt)
        n = C.c_size_t()
        p = L.emu_ticks(C.byref(n))
        tr = np.ctypeslib.as_array(p, shape=(n.value,)).copy().reshape(-1, a.lanes, 4)
        live = (tr[:, :, 3] >> 31).astype(bool)
        full = live.sum(axis=1) >= int(0.9 * a.lanes)              # the steady part: (nearly) every lane has a read
        tr = tr[full]; live = live[full]
        rounds = tr.shape[0]
        sweeps, iters, locus = tr[:, :, 0], tr[:, :, 1], tr[:, :, 2]
        nw = a.lanes // 64
        def per_wave(x):
            return x.reshape(rounds, nw, 64).max(axis=2).sum() / (rounds * nw)
        L.emu_again.restype = C.POINTER(C.c_uint64)
        ag = np.ctypeslib.as_array(L.emu_again(), shape=(64,)).copy()
        C.memset(L.emu_again(), 0, 64 * 8)
        import re
        src = open(os.path.join(ROOT, "bowtie_amd/csrc/bt_core.h")).read()
        m = re.search(r"enum \{\s*ST_IDLE = 0,(.*?)\};", src, re.S)
        items = ["ST_IDLE"] + [x.strip() for x in re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).replace("\n", " ").split(",") if x.strip()]
        print("   states a second or later sweep of one call starts in (per read):", {items[k]: round(float(ag[k]) / a.reads, 2) for k in np.argsort(-ag)[:8] if ag[k]})
        print("%s: %d full rounds of %d lanes; per wavefront and round: %.2f trips round the loop, %.2f sweeps, %.2f locus blocks  "
              "(per lane and round: %.2f / %.2f / %.2f); lane rounds per read %.1f" %
              (label, rounds, a.lanes, per_wave(iters), per_wave(sweeps), per_wave(locus), iters[live].mean(), sweeps[live].mean(), locus[live].mean(),
               cnt.lane_iters / a.reads))


if __name__ == "__main__":
    main()
