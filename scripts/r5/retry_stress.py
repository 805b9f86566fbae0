"""Stress of bt_align_batch's overflow second pass (round 5: in the sixth GPU call test_gpu_scratch_overflow_is_retried came
back ONCE, under six xdist workers sharing the GPU, with the right hits and wrong mismatch lists -- round 3's symptom
(DESIGN.md 4.3), this time not on the streamed path; alone it passes).  One process, many rounds of that test's five
cases with absurdly small arenas (so that a fifth of the reads take the second pass), every round's reads in an order of
its own; several of these side by side are the load.  For every read that differs from the oracle it says what differs:
the hit itself, or only the mismatch list -- and whether the read went through the second pass.

  python scripts/r5/retry_stress.py --seconds 60 --tag A
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BT_ENTRY_CAP", "12")
os.environ.setdefault("BT_FRAME_CAP", "3")
os.environ.setdefault("BT_PARTIAL_CAP", "4")

import common as T  # noqa: E402
from bowtie_amd import _abi as A  # noqa: E402
from bowtie_amd import aligner as AL  # noqa: E402
from bowtie_amd.reads import ReadBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    cases = [("multi", "syn100", "n2"), ("multi", "syn50lowq", "n3"), ("e_coli", "syn76", "v2"), ("multi", "syn76", "n1_a_m20"), ("multi", "syn36", "n2_k3")]
    gidx = {n: AL.Index(os.path.join(T.G, n)) for n in ("e_coli", "multi")}
    sets = {(i, r): T.read_set(i, r) for i, r, _ in cases}
    want = {c: T.oracle_results(c[0], sets[(c[0], c[1])], T.MODES[c[2]], cap=T.hit_cap_for(T.MODES[c[2]])) for c in cases}
    als = {c: AL.Aligner(gidx[c[0]], A.make_policy(**T.MODES[c[2]])) for c in cases}
    t_end = time.time() + a.seconds
    rounds = fails = 0
    reports = []
    while time.time() < t_end:
        rng = np.random.default_rng(7000003 * rounds + 11 + hash(a.tag) % 1000)
        for c in cases:
            b0 = sets[(c[0], c[1])]
            perm = rng.permutation(b0.n)
            b = ReadBatch(b0.seq[perm].copy(), b0.qual[perm].copy(), b0.len[perm].copy(), b0.seed[perm].copy(), [b0.names[i] for i in perm])
            kw = T.MODES[c[2]]
            got = als[c].align(b, hit_cap=T.hit_cap_for(kw))
            w = [want[c][i] for i in perm]
            bad = [i for i in range(b.n) if got[i] != w[i]]
            if bad:
                fails += 1
                only_mm = 0
                for i in bad:
                    g, x = got[i], w[i]
                    same_but_mm = g[1] == x[1] and g[2] == x[2] and len(g[0]) == len(x[0]) and all(
                        (h.tidx, h.toff, h.fw, h.cost, h.stratum, h.oms) == (y.tidx, y.toff, y.fw, y.cost, y.stratum, y.oms) for h, y in zip(g[0], x[0]))
                    only_mm += 1 if same_but_mm else 0
                if len(reports) < 4:
                    reports.append({"round": rounds, "case": list(c), "n": b.n, "n_bad": len(bad), "only_the_mismatch_list": only_mm,
                                    "retried": int(als[c].last_retried),
                                    "first": [{"i": int(i), "got": repr(got[i])[:300], "want": repr(w[i])[:300]} for i in bad[:2]]})
        rounds += 1
    print(json.dumps({"tag": a.tag, "rounds": rounds, "cases_per_round": len(cases), "fails": fails, "reports": reports,
                      "env": {k: os.environ.get(k) for k in ("BT_ENTRY_CAP", "BT_FRAME_CAP", "BT_LOCUS", "BT_MAX_BLOCKS", "BT_NO_RL")}}))


if __name__ == "__main__":
    main()
