#!/usr/bin/env python3
"""How many guarded state blocks does a wavefront execute per lock-step round -- and how many would it after re-dealing a
block's 256 reads to its four wavefronts by state every R rounds?  (the plan for bt_search_kernel.)

No GPU: the host build of the automaton (tests/emu) with a probe compiled in (a patched copy under /tmp, the product
header is not touched) records, per round and lane, which state blocks of bt_lane_run / bt_lane_slow the lane entered.
A wavefront's round executes the union of its lanes' blocks; the model counts them (every block one unit, the
request-issue-and-rank section one more, weighted by --top) for fixed wavefronts and for re-dealt ones.

  python scripts/divergence_model.py [--reads 6000] [--len 100] [--index tests/golden/e_coli]
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
W = "/tmp/bt_divergence_model"


def build_probe():
    os.makedirs(W + "/emu", exist_ok=True)
    s = open(os.path.join(ROOT, "bowtie_amd/csrc/bt_core.h")).read()
    s = s.replace("#ifndef BT_CORE_H_", 'extern "C" { extern unsigned long long* bt_visit_mask; }\n#define BT_VISIT(x) (*bt_visit_mask |= 1ull << (x))\n#ifndef BT_CORE_H_', 1)
    old = "#define ST_IS(x) (L.state == (x))"
    assert old in s
    s = s.replace(old, "#define ST_IS(x) (L.state == (x) && (BT_VISIT(x), true))")
    old = "#define ST_IS_NOREQ(x) (L.state == (x) && req.kind == RQ_NONE)"
    assert old in s
    s = s.replace(old, "#define ST_IS_NOREQ(x) (L.state == (x) && req.kind == RQ_NONE && (BT_VISIT(x), true))")
    for pat in ["\t\tif (!RL && L.state == ST_WIN_DONE) {", "\t\tif (L.state == ST_CHASE_LFDONE) {", "\t\tif (L.state == ST_STEP_LFDONE || L.state == ST_STEP_POST || (RL && L.state == ST_STEP_LOC)) {", "\t\tif (RL && L.state == ST_STEP_BEGIN && L.lmode) {",
                "\t\tif (L.state == ST_STEP_BEGIN) {", "\t\tif (L.state == ST_CHASE_CHECK) {"]:
        assert pat in s, pat
        s = s.replace(pat, pat + " BT_VISIT(L.state);", 1)
    s = s.replace("\twhile (BT_IS_SLOW(L.state) && req.kind == RQ_NONE) {", "\twhile (BT_IS_SLOW(L.state) && req.kind == RQ_NONE) { BT_VISIT(40);", 1)
    open(W + "/bt_core.h", "w").write(s)
    e = open(os.path.join(ROOT, "tests/emu/bt_emu.cpp")).read()
    e = e.replace('#include "../../bowtie_amd/csrc/bt_host.h"', '#include "../bt_host.h"')
    e = e.replace("/* Same contract as bt_align_batch (host pointers); nLanes lock-step lanes. */",
                  'extern "C" { unsigned long long* bt_visit_mask = nullptr; }\nstatic std::vector<unsigned long long> g_trace;\nstatic unsigned long long g_dummy;\n'
                  'extern "C" unsigned long long* emu_trace(size_t* n) { *n = g_trace.size(); return g_trace.data(); }\n')
    old = "\twhile (live > 0) {\n\t\tfor (uint32_t g = 0; g < nLanes; g++) {\n\t\t\tif (drained[g]) continue;"
    assert old in e
    e = e.replace(old, "\tg_trace.clear();\n\twhile (live > 0) {\n\t\tconst size_t row = g_trace.size();\n\t\tg_trace.resize(row + nLanes, 0ull);\n"
                       "\t\tfor (uint32_t g = 0; g < nLanes; g++) {\n\t\t\tbt_visit_mask = &g_trace[row + g];\n\t\t\tif (drained[g]) continue;")
    old = "\t\t\tif (drained[g]) continue;\n\t\t\tif (req.kind == RQ_FETCH && (L.state == ST_LOC_REC"
    assert old in e
    e = e.replace(old, "\t\t\tif (drained[g]) { g_trace[row + g] = 0; continue; }\n"
                       "\t\t\tg_trace[row + g] |= ((unsigned long long)(req.kind == RQ_RANK && req.n == 2) << 62) | (1ull << 63);\n\t\t\tif (req.kind == RQ_FETCH && (L.state == ST_LOC_REC")
    e = e.replace("\tout->mm_pool_used = mmUsed < out->mm_pool_cap ? mmUsed : out->mm_pool_cap;\n\tif (counts) {\n\t\tcounts->lfex = CNT[CN_LFEX];",
                  "\tbt_visit_mask = &g_dummy;\n\tout->mm_pool_used = mmUsed < out->mm_pool_cap ? mmUsed : out->mm_pool_cap;\n\tif (counts) {\n\t\tcounts->lfex = CNT[CN_LFEX];", 1)
    open(W + "/emu/bt_emu.cpp", "w").write(e)
    for f in ("bt_rank.h", "bt_best.h", "bt_host.h", "bt_host.cpp", "bt_kernels.h", "bt_io.h"):
        t = open(os.path.join(ROOT, "bowtie_amd/csrc", f)).read().replace("../../include/bowtie_amd.h", os.path.join(ROOT, "include/bowtie_amd.h"))
        open(os.path.join(W, f), "w").write(t)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w"] + os.environ.get("BT_EMU_DEFINES", "").split() +
                          ["-o", W + "/libbt_emu_dv.so", W + "/emu/bt_emu.cpp", W + "/bt_host.cpp"])     # e.g. BT_EMU_DEFINES=-DBT_LOCAL_LOOPS=1


def popcount(a):
    a = a.astype(np.uint64)
    c = np.zeros(a.shape, dtype=np.int64)
    for k in range(48):
        c += ((a >> np.uint64(k)) & np.uint64(1)).astype(np.int64)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--index", default=os.path.join(ROOT, "tests/golden/e_coli"))
    ap.add_argument("--synthetic", type=int, default=0, help="index a synthetic genome of this many bases (the benchmark's generator) instead")
    ap.add_argument("--top", type=float, default=6.0, help="cost of the request-issue-and-rank section in state-block units")
    a = ap.parse_args()
    build_probe()
    import emu_lib as E
    E.LIB_PATH = W + "/libbt_emu_dv.so"
    E.SRCS = []
    from bowtie_amd import _abi as A
    from bowtie_amd.synth import synth_reads
    import common as T
    if a.synthetic:
        # the benchmark's synthetic genome (repeat families and all) at a size the CPU indexes in a minute
        import torch
        from bowtie_amd import ebwt_build as EB
        a.index, text, note = EB.ensure_big_index(a.synthetic, torch.device("cpu"), cache_dir=W)
        print(note)
    else:
        text = T.joined_text(os.path.basename(a.index)) if os.path.basename(a.index) in ("e_coli", "multi") else None
        assert text is not None, "give one of the fixture indexes, or --synthetic <bp>"
    batch = synth_reads(text, a.reads, a.len, mm_dist=(0, 1, 2, 2, 3, 4), seed=11)
    L = E.lib()
    L.emu_trace.restype = C.POINTER(C.c_uint64)
    L.emu_trace.argtypes = [C.POINTER(C.c_size_t)]
    emu = E.EmuAligner(a.index)
    emu.align(A.make_policy(mode="n", mms=2), batch, n_lanes=256, lite=True, pal_cap=16384)
    n = C.c_size_t()
    p = L.emu_trace(C.byref(n))
    tr = np.ctypeslib.as_array(p, shape=(n.value,)).copy().reshape(-1, 256)
    live = (tr >> np.uint64(63)).astype(bool)
    hasb = ((tr >> np.uint64(62)) & np.uint64(1)).astype(bool)
    mask = tr & np.uint64((1 << 48) - 1)
    rounds = tr.shape[0]
    print("reads %d x %d bp on %s: %d rounds of a 256-lane block, %.1f lane-rounds per read, mean live lanes %.1f of 256"
          % (a.reads, a.len, os.path.basename(a.index), rounds, live.sum() / a.reads, live.sum() / rounds))

    def cost(assign):
        """assign[r] = permutation of the 256 lanes into 4 wavefronts of 64 for round r"""
        blocks = 0; tops = 0.0; waves = 0
        for r in range(rounds):
            m = mask[r][assign[r]].reshape(4, 64)
            lv = live[r][assign[r]].reshape(4, 64)
            hb = hasb[r][assign[r]].reshape(4, 64)
            for w in range(4):
                if not lv[w].any():
                    continue
                waves += 1
                u = np.bitwise_or.reduce(m[w])
                blocks += int(popcount(np.array([u]))[0])
                tops += a.top * (1.0 if hb[w].any() else 0.55)      # the second row's rank is skipped when no lane needs it
        return blocks, tops, waves

    ident = np.arange(256)
    base = cost([ident] * rounds)
    print("fixed wavefronts:           %.2f state blocks per wave-round (+ %.2f for the issue/rank section), %d wave-rounds"
          % (base[0] / base[2], base[1] / base[2], base[2]))
    # the state a lane is in at the start of a round = lowest set bit class is not recorded; use the visit mask of the
    # previous round's last state as the key: approximate by the lowest block it enters this round
    def key_of(r):
        m = mask[r]
        k = np.full(256, 63, dtype=np.int64)
        for b in range(47, -1, -1):
            k = np.where((m >> np.uint64(b)) & np.uint64(1), b, k)
        k = np.where(live[r], k, 99)
        return k
    for R in (1, 4, 16, 64):
        assign = []
        cur = ident
        for r in range(rounds):
            if r % R == 0:
                cur = np.argsort(key_of(r), kind="stable")
            assign.append(cur)
        c = cost(assign)
        print("re-dealt every %2d rounds:   %.2f state blocks per wave-round (+ %.2f), %d wave-rounds; total cost %.2f of fixed"
              % (R, c[0] / c[2], c[1] / c[2], c[2], (c[0] + c[1]) / (base[0] + base[1])))

    defer_model(mask, live, hasb, a.top)


def defer_model(mask, live, hasb, top):
    """The other candidate: run the slow-state sweep only every k-th round (or once enough lanes wait for it).  A lane
    whose step needs the sweep in a round without one consumes its request's answer (the fast part of its step) and waits.
    Lanes' step sequences are taken from the recorded run; fixed wavefronts."""
    import re
    src = open(os.path.join(ROOT, "bowtie_amd/csrc/bt_core.h")).read()
    m = re.search(r"enum \{\s*ST_IDLE = 0,(.*?)\};", src, re.S)
    items = [x.strip() for x in re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).replace("\n", " ").split(",") if x.strip()]
    idx = {nm: i + 1 for i, nm in enumerate(items)}
    PRE = sum(1 << idx[n] for n in ("ST_STEP_LFDONE", "ST_STEP_POST", "ST_CHASE_LFDONE", "ST_WIN_DONE"))
    SWEEP = 1 << 40
    seqs = []
    for g in range(256):
        rows = np.nonzero(live[:, g])[0]
        seqs.append([(int(mask[r, g]), bool(hasb[r, g])) for r in rows])

    def simulate(k, thresh):
        total = 0.0; waverounds = 0
        for w in range(4):
            lanes = list(range(64 * w, 64 * w + 64))
            ptr = {g: 0 for g in lanes}; pend = {}
            r = 0
            while True:
                act = [g for g in lanes if ptr[g] < len(seqs[g]) or g in pend]
                if not act:
                    break
                need = [g for g in act if (g in pend) or (seqs[g][ptr[g]][0] & SWEEP)]
                run_sweep = (r % k == 0) or (len(need) >= thresh)
                u = 0; hb = False
                for g in act:
                    if g in pend:
                        if run_sweep:
                            u |= pend.pop(g)
                        continue
                    mk, b = seqs[g][ptr[g]]
                    ptr[g] += 1; hb |= b
                    if (mk & SWEEP) and not run_sweep:
                        u |= mk & PRE; pend[g] = mk & ~PRE
                    else:
                        u |= mk
                total += top * (1.0 if hb else 0.55) + bin(u).count("1")
                waverounds += 1; r += 1
        return total, waverounds
    base = simulate(1, 0)
    print("slow sweep every round (as now): %.2f units per wave-round" % (base[0] / base[1]))
    for k, th in ((2, 99), (3, 24), (4, 24), (8, 24)):
        c = simulate(k, th)
        print("sweep every %d rounds or when >= %2d lanes wait: total cost %.2f of now, wave-rounds %.2f of now" % (k, th, c[0] / base[0], c[1] / base[1]))


if __name__ == "__main__":
    main()
