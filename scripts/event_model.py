#!/usr/bin/env python3
"""Where does a wavefront wait INSIDE a lock-step round?  The automaton's rule is one memory request per lane per round,
answered at the top of the next; the places where a state block still loads from global memory on the spot (frame
mismatch words, a seedling, the fragment table's binary search, a read's chunks at pick-up) stall the whole wavefront
for a memory latency each.  No GPU: the host build of the automaton with probes at those sites (a patched copy under
/tmp) records, per round and lane, which of them the lane went through; a wavefront's round pays for the union.

  python scripts/event_model.py [--reads 3000] [--len 100] [--synthetic 6000000 | --index tests/golden/e_coli]
"""
import argparse, ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import divergence_model as DM

SITES = [  # (bit, name, unique source text to put the probe after)
    (41, "lane_start: len/seed + read chunks", "\tL.rd = rd;\n"),
    (42, "lane_finish: batch descriptor", "\tBT_GP(uint32_t, B.n_hits)[L.rd] = L.nhits;\n"),
    (43, "report_hit: pool cursor, frame words", "\tL.nhits++;\n\tif (L.nhits > P.sinkMax) return true;\n"),
    (44, "RESOLVE_DONE: rstarts binary search", "\t\t\tconst uint32_t* rstarts = IXSEL(rstarts);\n"),
    (45, "RA_BEGIN: calcStratum frame words", "\t\t\t\tif ((FRW(i, FR_MM) & 0xffffu) >= (L.qlen - L.r3)) stratum++;\n"),
    (46, "SEARCH_END: seedling", "\t\t\t\tconst uint64_t pal = PALS(L.palIdx);\n"),
    (47, "STEP_LFDONE: half-and-half frame words", "\t\t\t\t\t\tuint32_t dd = L.qlen - (FRW(i, FR_MM) & 0xffffu) - 1u;\n\t\t\t\t\t\tif (dd < L.d5) hi++;"),
    (48, "hh_check_top: frame words", "\t\t\t\tuint32_t dd = L.qlen - (FRW(i, FR_MM) & 0xffffu) - 1u;\n\t\t\t\tif (dd >= L.d5 && dd < L.d3) lo++;"),
    (49, "report_partial: frame words", "\tuint64_t p0 = 0xffff, p1 = 0xffff, p2 = 0xffff, c0 = 3, c1 = 3, c2 = 3;\n"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=3000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--index", default=os.path.join(ROOT, "tests/golden/e_coli"))
    ap.add_argument("--synthetic", type=int, default=0)
    a = ap.parse_args()
    DM.build_probe()
    p = DM.W + "/bt_core.h"
    s = open(p).read()
    for bit, name, pat in SITES:
        assert s.count(pat) == 1, (name, s.count(pat))
        if pat.lstrip().startswith("if ((FRW"):
            s = s.replace(pat, pat.replace("if ((FRW(i, FR_MM) & 0xffffu)", "if ((BT_VISIT(%d), (FRW(i, FR_MM) & 0xffffu))" % bit))
        elif pat.lstrip().startswith("uint32_t dd"):
            s = s.replace(pat, ("BT_VISIT(%d);\n" % bit) + pat)
        else:
            s = s.replace(pat, pat + " BT_VISIT(%d);\n" % bit)
    open(p, "w").write(s)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", DM.W + "/libbt_emu_dv.so", DM.W + "/emu/bt_emu.cpp", DM.W + "/bt_host.cpp"])
    import emu_lib as E
    E.LIB_PATH = DM.W + "/libbt_emu_dv.so"; E.SRCS = []
    from bowtie_amd import _abi as A
    from bowtie_amd.synth import synth_reads
    import common as T
    if a.synthetic:
        import torch
        from bowtie_amd import ebwt_build as EB
        a.index, text, note = EB.ensure_big_index(a.synthetic, torch.device("cpu"), cache_dir=DM.W)
        print(note)
    else:
        text = T.joined_text(os.path.basename(a.index))
    batch = synth_reads(text, a.reads, a.len, mm_dist=(0, 1, 2, 2, 3, 4), seed=11)
    L = E.lib()
    L.emu_trace.restype = C.POINTER(C.c_uint64); L.emu_trace.argtypes = [C.POINTER(C.c_size_t)]
    E.EmuAligner(a.index).align(A.make_policy(mode="n", mms=2), batch, n_lanes=256, lite=True, pal_cap=16384)
    n = C.c_size_t(); ptr = L.emu_trace(C.byref(n))
    tr = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy().reshape(-1, 256)
    live = (tr >> np.uint64(63)).astype(bool)
    rounds = tr.shape[0]
    waves = 0
    for w in range(4):
        waves += int(live[:, 64 * w:64 * w + 64].any(axis=1).sum())
    print("reads %d x %d bp: %.1f lane-rounds per read, %d wave-rounds" % (a.reads, a.len, live.sum() / a.reads, waves))
    print("%-44s %10s %14s" % ("site (loads on the spot)", "per read", "of wave-rounds"))
    tot = np.zeros((rounds, 4), dtype=np.int64)
    for bit, name, _ in SITES:
        m = ((tr >> np.uint64(bit)) & np.uint64(1)).astype(bool)
        wr = 0
        for w in range(4):
            hit = m[:, 64 * w:64 * w + 64].any(axis=1)
            wr += int(hit.sum()); tot[:, w] += hit
        print("%-44s %10.2f %13.1f%%" % (name, m.sum() / a.reads, 100.0 * wr / waves))
    print("%-44s %10s %14.2f" % ("sites stalling a wavefront, per wave-round", "", tot.sum() / waves))


main()
