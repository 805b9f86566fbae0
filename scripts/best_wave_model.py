#!/usr/bin/env python3
"""How a wavefront's 64 lanes are used by bt_best_kernel's loop -- the wavefront automaton of bt_best.h -- under given
gate settings: the host build of the engine (tests/emu) goes through the kernel's loop with 64 lanes side by side and
counts hot rounds, the lanes in each piece of a round, cold sweeps and the lanes in them.  No GPU; e_coli workloads of
bench.py's shape.  A cost line prices the pieces with the wavefront cycles of round 4's section profile
(profiles/r4/fourth_call_best_sections.txt), which is what picked the gates' defaults before the GPU A/B.

  python scripts/best_wave_model.py [--workload n2_best|pe] [--reads 3000] [--gates 16/16/1/8,32/16/4/24,...]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import emu_lib, common as T
from bowtie_amd import _abi as A
from bowtie_amd.synth import synth_reads, synth_pairs
em = emu_lib.EmuAligner(os.path.join(T.ROOT, "tests", "golden", "e_coli"))
text = T.joined_text("e_coli")
if %(workload)r == "pe":
    pol = A.make_policy(mode="n", mms=1, best=True, max_ins=500)
    b1, b2 = synth_pairs(text, %(reads)d // 2, 50, seed=7)[:2]
    res = em.align_pairs(pol, b1, b2, hit_cap=2)
else:
    kw = T.MODES[%(workload)r]
    pol = A.make_policy(**kw)
    b = synth_reads(text, %(reads)d, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=99)
    res = em.align(pol, b, hit_cap=T.hit_cap_for(kw))
'''

# wavefront cycles per pass of a piece, from the section profile of the call-by-call kernel on e_coli -n 2 --best (s_memtime
# units; a streak pass there is ~20 steps long): a step, the end of a streak, a piece of a walk, a cold sweep (a driver's
# second + first halves, the runner's turn, the sort), a read's begin
COST = dict(step=22e3, send=173e3, chase=10e3, sweep=575e3, sweep2=300e3, take=3.2e6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="n2_best")
    ap.add_argument("--reads", type=int, default=3000)
    ap.add_argument("--lanes", type=int, default=64)
    ap.add_argument("--gates", default="16/16/1/8,8/16/1/8,32/16/1/8,16/16/2/16,16/16/4/24,32/16/4/24,32/32/4/24,48/32/4/32,24/16/3/16")
    a = ap.parse_args()
    for g in a.gates.split(","):
        c, t, sp, sm = g.split("/")
        env = dict(os.environ, BT_EMU_VERBOSE="1", BT_EMU_WAVE_LANES=str(a.lanes), BT_BEST_COLD_MIN=c, BT_BEST_TAKE_MIN=t,
                   BT_BEST_SEND_PERIOD=sp, BT_BEST_SEND_MIN=sm)
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, workload=a.workload, reads=a.reads)], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        line = [l for l in p.stderr.decode().splitlines() if l.startswith("[emu] automaton")]
        if not line:
            print(g, "no result", p.stderr.decode()[-400:])
            continue
        l = line[-1]
        m = re.search(r"(\d+) hot rounds \(step (\d+) x ([\d.]+) lanes, send (\d+) x ([\d.]+), chase (\d+) x ([\d.]+); idle lane-rounds (\d+)\), (\d+) cold sweeps x ([\d.]+) lanes \(second pass (\d+) x ([\d.]+); reads taken in (\d+) x ([\d.]+)\)", l)
        hot, stepR, stepL, sendR, sendL, chR, chL, idle, sw, swL, sw2, sw2L, tk, tkL = [float(x) for x in m.groups()]
        cost = stepR * COST["step"] + sendR * COST["send"] + chR * COST["chase"] + sw * COST["sweep"] + sw2 * COST["sweep2"] + tk * COST["take"]
        print("gates cold/take/sendPeriod/sendMin %-12s hot %6d (step %6d x %4.1f, send %6d x %4.1f, chase %5d x %4.1f, idle %4.1f%%) cold %5d x %4.1f (2nd %5d x %4.1f) takes %4d x %4.1f  modelled %.2f Gcycles" %
              (g, hot, stepR, stepL, sendR, sendL, chR, chL, 100.0 * idle / max(1.0, hot * a.lanes), sw, swL, sw2, sw2L, tk, tkL, cost / 1e9))


if __name__ == "__main__":
    main()
