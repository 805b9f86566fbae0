#!/usr/bin/env python3
"""How many lock-step rounds of bt_search_kernel are spent on ranges that are a single BWT row -- and what would a read
cost if the automaton left row space there ("locus mode": one dense-SA look-up gives the text position, the rest of that
subtree is string comparison against the 2-bit text)?   Round 5, VERDICT item 1: measure before building.

No GPU: the host build of the automaton (tests/emu) with an event probe compiled in (a patched copy under /tmp, the product
header is not touched) records every lane-round's request.  For single-row rank requests the probe walks the suffix array
on the host, so every such request is labelled with its anchor (text position + depth), which is what identifies the
locus a subtree compares against.

  python scripts/textmode_model.py [--reads 4000] [--len 100] [--synthetic 30000000] [--mode n2|v2]
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
W = "/tmp/bt_textmode_model"


def build_probe():
    os.makedirs(W + "/emu", exist_ok=True)
    e = open(os.path.join(ROOT, "tests/emu/bt_emu.cpp")).read()
    e = e.replace('#include "../../bowtie_amd/csrc/bt_host.h"', '#include "../bt_host.h"')
    e = e.replace("/* Same contract as bt_align_batch (host pointers); nLanes lock-step lanes. */",
                  'struct Ev { uint32_t rd, code, anchor, d, sd, lane, fr; };\nstatic std::vector<Ev> g_ev;\n'
                  'extern "C" Ev* emu_events(size_t* n) { *n = g_ev.size(); return g_ev.data(); }\n'
                  'static uint32_t sa_of(const BtIndexDev& ix, uint32_t row) { uint32_t j = 0; while ((row & ix.offMask) != row && row != ix.zOff) { uint32_t lf[4], l; bt_rank4(ix, row, lf, &l); row = lf[l]; j++; } return (row == ix.zOff ? 0u : ix.offs[row >> ix.offRate]) + j; }\n')
    old = "\t\t\tif (req.kind == RQ_RANK) {\n\t\t\t\tif (L.lfk == LFK_CHASE) BT_COUNT(CN_CHASE);"
    assert old in e
    e = e.replace(old, "\t\t\t{ Ev v; v.rd = L.rd; v.lane = g; v.d = L.d; v.sd = L.sd; v.anchor = 0xffffffffu; v.fr = (L.lmode ? (uint32_t)L.dcf - L.depth : L.d - L.depth + 1u) | ((uint32_t)L.lmode << 16) | ((uint32_t)L.altNum << 20);\n"
                       "\t\t\t  if (req.kind == RQ_RANK) { const bool single = L.lfk == LFK_LF1 || L.lfk == LFK_CHASE || (req.n == 2 && (uint32_t)req.x == (uint32_t)req.a + 1u);\n"
                       "\t\t\t    v.code = (L.lfk == LFK_EX2 ? 1u : L.lfk == LFK_C2 ? 2u : L.lfk == LFK_LF1 ? 3u : 4u) | (single ? 8u : 0u) | ((uint32_t)L.mirror << 4);\n"
                       "\t\t\t    if (single && L.lfk != LFK_CHASE) v.anchor = sa_of(e->d[L.mirror], (uint32_t)req.a) + L.d; }\n"
                       "\t\t\t  else v.code = 5u | ((uint32_t)(L.state) << 8);\n"
                       "\t\t\t  g_ev.push_back(v); }\n" + old, 1)
    open(W + "/emu/bt_emu.cpp", "w").write(e)
    for f in ("bt_core.h", "bt_rank.h", "bt_best.h", "bt_host.h", "bt_host.cpp", "bt_kernels.h", "bt_io.h"):
        t = open(os.path.join(ROOT, "bowtie_amd/csrc", f)).read().replace("../../include/bowtie_amd.h", os.path.join(ROOT, "include/bowtie_amd.h"))
        open(os.path.join(W, f), "w").write(t)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", W + "/libbt_emu_tm.so", W + "/emu/bt_emu.cpp", W + "/bt_host.cpp"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--index", default=os.path.join(ROOT, "tests/golden/e_coli"))
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--mode", default="n2")
    a = ap.parse_args()
    build_probe()
    import emu_lib as E
    E.LIB_PATH = W + "/libbt_emu_tm.so"
    E.SRCS = []
    from bowtie_amd import _abi as A
    from bowtie_amd.synth import synth_reads
    import common as T
    if a.synthetic:
        import torch
        from bowtie_amd import ebwt_build as EB
        a.index, text, note = EB.ensure_big_index(a.synthetic, torch.device("cpu"), cache_dir=W)
        print(note)
    else:
        text = T.joined_text(os.path.basename(a.index))
    mmd = (0, 1, 2, 2, 3, 4) if a.len >= 100 else (0, 0, 1, 1, 2, 3)
    batch = synth_reads(text, a.reads, a.len, mm_dist=mmd, seed=11)
    L = E.lib()

    class Ev(C.Structure):
        _fields_ = [(k, C.c_uint32) for k in ("rd", "code", "anchor", "d", "sd", "lane", "fr")]
    L.emu_events.restype = C.POINTER(Ev)
    L.emu_events.argtypes = [C.POINTER(C.c_size_t)]
    emu = E.EmuAligner(a.index)
    pol = A.make_policy(mode="n", mms=2) if a.mode == "n2" else A.make_policy(mode="v", mms=2)
    cnt = A.OpCounts()
    emu.align(pol, batch, n_lanes=256, lite=True, pal_cap=16384, counts=cnt)
    n = C.c_size_t()
    p = L.emu_events(C.byref(n))
    ev = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 7)).copy()
    R = a.reads
    print("%d reads x %d bp, %s: %.1f lane-rounds per read" % (R, a.len, a.mode, len(ev) / R))
    code = ev[:, 1] & 7
    single = (ev[:, 1] & 8) != 0
    for nm, c in (("mapLFEx", 1), ("mapLF x2", 2), ("mapLF1", 3), ("SA walk", 4), ("fetch", 5)):
        m = code == c
        print("  %-9s %7.1f per read, of which on a single row %7.1f" % (nm, m.sum() / R, (m & single).sum() / R))
    # fetch rounds by the state the lane is left in
    import re
    src = open(os.path.join(ROOT, "bowtie_amd/csrc/bt_core.h")).read()
    m = re.search(r"enum \{\s*ST_IDLE = 0,(.*?)\};", src, re.S)
    items = ["ST_IDLE"] + [x.strip() for x in re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).replace("\n", " ").split(",") if x.strip()]
    fm = code == 5
    st = ev[fm, 1] >> 8
    frl = ev[fm, 6] & 0xffff
    alt = ev[fm, 6] >> 20
    for s in np.unique(st):
        m2 = st == s
        print("    fetch -> %-18s %6.1f per read; frame's row-space entries: mean %.1f, <=4: %.2f <=8: %.2f <=16: %.2f; altNum mean %.1f <=4: %.2f" % (items[s], m2.sum() / R, frl[m2].mean(), (frl[m2] <= 4).mean(), (frl[m2] <= 8).mean(), (frl[m2] <= 16).mean(), alt[m2].mean(), (alt[m2] <= 4).mean()))
    # per read: order events by (lane, time) -- events are appended round by round, lanes in order; a read stays on its lane
    order = np.lexsort((np.arange(len(ev)), ev[:, 0]))
    ev = ev[order]
    code = ev[:, 1] & 7
    single = (ev[:, 1] & 8) != 0
    rd = ev[:, 0]
    anchor = ev[:, 2].astype(np.int64) | ((ev[:, 1].astype(np.int64) >> 4 & 1) << 40)
    # LF1 streaks: maximal runs of mapLF1 requests of a read
    is1 = code == 3
    brk = np.ones(len(ev), bool)
    brk[1:] = (rd[1:] != rd[:-1]) | (is1[1:] != is1[:-1])
    run_id = np.cumsum(brk) - 1
    lens = np.bincount(run_id[is1])
    lens = lens[lens > 0]
    tot = lens.sum()
    print("mapLF1 streaks: %d runs, mean length %.1f; share of mapLF1 steps in streaks of length >= 4 / 8 / 16 / 32: %.2f %.2f %.2f %.2f"
          % (len(lens), lens.mean(), lens[lens >= 4].sum() / tot, lens[lens >= 8].sum() / tot, lens[lens >= 16].sum() / tot, lens[lens >= 32].sum() / tot))
    # locus mode: every single-row rank request disappears; a new anchor costs 2 rounds (dense SA, text window), an anchor
    # seen before by this read but not the cached one costs 1 (text window again); SA walks become one look-up per reported row
    srow = single & (code != 4)
    new_anchor = 0; revisit = 0
    cur_rd = -1; cached = None; seen = set()
    a_list = anchor[srow]; r_list = rd[srow]
    for k in range(len(a_list)):
        if r_list[k] != cur_rd:
            cur_rd = r_list[k]; cached = None; seen = set()
        x = a_list[k]
        if x != cached:
            if x in seen: revisit += 1
            else: new_anchor += 1; seen.add(x)
            cached = x
    walks = code == 4
    wbrk = np.ones(len(ev), bool)
    wbrk[1:] = (rd[1:] != rd[:-1]) | (walks[1:] != walks[:-1])
    n_walks = int((wbrk & walks).sum())
    old = len(ev)
    new = old - int(srow.sum()) - int(walks.sum()) + 2 * new_anchor + revisit
    print("locus mode: %.1f single-row rank rounds and %.1f SA-walk rounds per read go; %.2f new anchors (x2 rounds) and %.2f revisits (x1) per read come"
          % (srow.sum() / R, walks.sum() / R, new_anchor / R, revisit / R))
    print("            rounds per read %.1f -> %.1f (%.2f of before); %d SA walks of mean %.1f steps" % (old / R, new / R, new / old, n_walks, walks.sum() / max(1, n_walks)))
    # heavy tail
    per_read_old = np.bincount(rd, minlength=R)
    print("            per-read rounds: median %d, mean %.1f, p99 %d, max %d" % (np.median(per_read_old), per_read_old.mean(), np.percentile(per_read_old, 99), per_read_old.max()))
    print("op counts of the run: lfex %.1f lf2 %.1f lf1 %.1f chase %.1f frames %.1f fetches %.1f per read"
          % (cnt.lfex / R, cnt.lf2 / R, cnt.lf1 / R, cnt.chase / R, cnt.frames / R, cnt.fetches / R))


if __name__ == "__main__":
    main()
