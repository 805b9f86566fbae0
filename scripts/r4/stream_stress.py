"""Stress of bt_align_stream_submit/_collect (the rare wrong mismatch list of round 3).

One process, many rounds; every round builds a fresh context and streams the six host batches of
tests/test_gpu_parity.py::test_gpu_host_batches_streamed through it with a carry-over drawn from
{0, 1, 12} -- but with every batch's reads in an order of its own (seeded by the round), so that what a
recycled staging area still holds from the round before is never the right answer for this one.  The oracle's
results are computed once per read set and permuted alongside.  Several of these processes side by side are
the "load" under which the failure was seen.

  python scripts/r4/stream_stress.py --seconds 120 --tag A [--carry 12]
env: BT_STREAM_POISON / BT_STREAM_ORDERED / BT_STREAM_RECHECK (bt_api.cpp), BT_MAX_BLOCKS (default 2 here)
Prints one JSON line: rounds, failures, and for the first failures what differed.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BT_MAX_BLOCKS", "2")

import common as T  # noqa: E402
from bowtie_amd import _abi as A  # noqa: E402
from bowtie_amd import aligner as AL  # noqa: E402
from bowtie_amd.reads import ReadBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--tag", default="")
    ap.add_argument("--carry", type=int, default=-1, help="-1: cycle through 0, 1, 12, 12")
    ap.add_argument("--mode", default="n2_k3")
    ap.add_argument("--no-permute", action="store_true")
    a = ap.parse_args()
    if os.environ.get("STRESS_NO_PERMUTE"):
        a.no_permute = True
    kw = T.MODES[a.mode]
    cap = 8
    names = ["syn100", "syn36", "syn150", "syn50lowq", "syn76", "syn100"]
    sets = {r: T.read_set("multi", r) for r in set(names)}
    want = {r: T.oracle_results("multi", sets[r], kw, cap=cap) for r in sets}
    gidx = AL.Index(os.path.join(T.G, "multi"))
    L = AL.lib()
    pol = A.make_policy(**kw)
    t_end = time.time() + a.seconds
    rounds = fails = 0
    reports = []
    cyc = [0, 1, 12, 12]
    while time.time() < t_end:
        carry = a.carry if a.carry >= 0 else cyc[rounds % len(cyc)]
        rng = np.random.default_rng(1000003 * rounds + 17)
        al = AL.Aligner(gidx, pol)
        assert L.bt_ctx_set_carry(al._h, carry) == 0
        jobs = []
        for r in names:
            b0 = sets[r]
            perm = np.arange(b0.n) if a.no_permute else rng.permutation(b0.n)
            b = ReadBatch(b0.seq[perm].copy(), b0.qual[perm].copy(), b0.len[perm].copy(), b0.seed[perm].copy(), [b0.names[i] for i in perm])
            k, rb = AL.pack_batch(b)
            hits = np.zeros(b.n * cap, dtype=A.HIT_DTYPE)
            n_hits = np.zeros(b.n, dtype=np.uint32)
            status = np.zeros(b.n, dtype=np.uint8)
            pool = np.zeros(b.n * cap * 8, dtype=np.uint16)
            hb = A.HitBatchC(cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
            jobs.append(dict(r=r, b=b, perm=perm, keep=k, rb=rb, hits=hits, n_hits=n_hits, status=status, pool=pool, hb=hb))
        done = []
        tag = C.c_void_p()
        for i, j in enumerate(jobs):
            rc = L.bt_align_stream_submit(al._h, C.byref(j["rb"]), C.byref(j["hb"]), C.c_void_p(i + 1))
            assert rc == 0, rc
            while True:
                rc = L.bt_align_stream_collect(al._h, C.byref(tag), 0)
                assert rc == 0, rc
                if tag.value is None:
                    break
                done.append(tag.value)
        while True:
            rc = L.bt_align_stream_collect(al._h, C.byref(tag), 1)
            assert rc == 0, rc
            if tag.value is None:
                break
            done.append(tag.value)
        ok = done == list(range(1, len(jobs) + 1))
        what = [] if ok else ["collect order %r" % done]
        for bi, j in enumerate(jobs):
            got = AL.unpack_hits(j["b"].n, cap, j["hits"], j["n_hits"], j["status"], j["pool"], int(pol.khits), int(pol.mhits), bool(pol.all_hits))
            w = [want[j["r"]][int(p)] for p in j["perm"]]
            bad = [i for i in range(j["b"].n) if got[i] != w[i]]
            if bad:
                ok = False
                raw = j["hits"].reshape(j["b"].n, cap)
                kinds = {"hit": 0, "mm_only": 0, "count": 0, "poison": 0}
                for i in bad:
                    g, ww = got[i], w[i]
                    if g[1] != ww[1] or g[2] != ww[2] or len(g[0]) != len(ww[0]):
                        kinds["count"] += 1
                    elif any((x.tidx, x.toff, x.fw, x.cost, x.stratum) != (y.tidx, y.toff, y.fw, y.cost, y.stratum) for x, y in zip(g[0], ww[0])):
                        kinds["hit"] += 1
                    else:
                        kinds["mm_only"] += 1
                    if any(p == 0x3ff for x in g[0] for p, _ in x.mms):
                        kinds["poison"] += 1
                first = bad[:4]
                what.append(dict(batch=bi + 1, reads=j["r"], carry=carry, n_bad=len(bad), kinds=kinds, mm_pool_used=int(j["hb"].mm_pool_used),
                                 first=[dict(i=i, mm=[(int(h["mm_off"]), int(h["nmm"])) for h in raw[i][:3]],
                                             pool=[hex(int(x)) for x in j["pool"][int(raw[i][0]["mm_off"]):int(raw[i][0]["mm_off"]) + 3]],
                                             got=repr(got[i]), want=repr(w[i])) for i in first],
                                 bad_idx=bad[:80]))
        rounds += 1
        if not ok:
            fails += 1
            if len(reports) < 5:
                reports.append(dict(round=rounds - 1, carry=carry, what=what))
        del al
    print(json.dumps(dict(tag=a.tag, rounds=rounds, fails=fails, env={k: os.environ.get(k) for k in ("BT_STREAM_POISON", "BT_STREAM_ORDERED", "BT_STREAM_RECHECK", "BT_STREAM_OLD_CURSOR", "STRESS_NO_PERMUTE", "BT_MAX_BLOCKS", "BT_LIB")}, reports=reports)))


if __name__ == "__main__":
    main()
