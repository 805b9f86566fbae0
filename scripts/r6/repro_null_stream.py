"""The wrong-mismatch-list bug of rounds 3-5, made deterministic.

  BT_LIB=libbowtie_amd_parent.so python scripts/r6/repro_null_stream.py     # the parent of the fix: FAILS
  python scripts/r6/repro_null_stream.py                                     # the tree: passes

1. Probe: how long hipMemset / hipMemcpy on the null stream take ON THE HOST while the null stream is busy for 200 ms
   (tests/emu/gpu_stall.hip).  hipMemset returning in microseconds = the fill is merely enqueued.
2. The overflow test's cases with the null stream held busy while each context is created and runs its first batch
   (what tests/test_gpu_parity.py::test_gpu_fresh_context_is_ordered_on_its_own_stream does), with a classification of
   every read that differs: did it take the second pass, where do its mismatch entries lie relative to the first pass's cursor.
Prints one JSON line; exit code 1 if any read differs."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BT_ENTRY_CAP", "12")
os.environ.setdefault("BT_FRAME_CAP", "3")
os.environ.setdefault("BT_PARTIAL_CAP", "4")

import common as T  # noqa: E402
import emu_lib  # noqa: E402
from bowtie_amd import _abi as A  # noqa: E402
from bowtie_amd import aligner as AL  # noqa: E402


def main():
    import torch
    stall = emu_lib.stall_lib()
    out = {"lib": os.path.basename(AL.LIB_PATH), "stall_ms": int(os.environ.get("STALL_MS", "250"))}
    torch.zeros(1, device="cuda")
    out["probe_us"] = {"hipMemset_64B": int(stall.gpu_stall_probe(0, 200, 64)), "hipMemset_64MB": int(stall.gpu_stall_probe(0, 200, 64 << 20)),
                       "hipMemcpy_h2d_64B": int(stall.gpu_stall_probe(1, 200, 64))}
    gidx = {n: AL.Index(os.path.join(T.G, n)) for n in ("e_coli", "multi")}
    keep = [AL.Aligner(gidx[ix], A.make_policy(**T.MODES["n2"])) for ix in ("e_coli", "multi")]
    torch.cuda.synchronize()
    emu = {n: emu_lib.EmuAligner(os.path.join(T.G, n)) for n in ("e_coli", "multi")}
    cases = []
    n_bad_total = 0
    for rep in range(int(os.environ.get("REPS", "3"))):
        for index, rname, mode in (("multi", "syn100", "n2"), ("multi", "syn50lowq", "n3"), ("e_coli", "syn76", "v2"), ("multi", "syn76", "n1_a_m20"), ("multi", "syn36", "n2_k3")):
            batch = T.read_set(index, rname)
            kw = T.MODES[mode]
            cap = T.hit_cap_for(kw)
            want = T.oracle_results(index, batch, kw, cap=cap)
            # which reads outgrow the tiny arenas (the second pass's reads): the host build of the automaton says
            flagged = [bool(r[2] & A.BT_ST_OVERFLOW) for r in emu[index].align(A.make_policy(**kw), batch, hit_cap=cap, fr_cap=3, ent_cap=12, pal_cap=4, lite=batch.stride <= 104)]
            if out["stall_ms"]:
                assert stall.gpu_stall(None, out["stall_ms"]) == 0
            al = AL.Aligner(gidx[index], A.make_policy(**kw))
            keep.append(al)
            rec = {"rep": rep, "case": "%s %s %s" % (index, rname, mode), "n": batch.n, "second_pass_reads": sum(flagged)}
            try:
                got = al.align(batch, hit_cap=cap)
            except AL.BowtieAmdError as e:
                rec["error"] = str(e)
                n_bad_total += 1
                cases.append(rec)
                torch.cuda.synchronize()
                continue
            rec["first_pass_cursor"] = int(AL.lib().bt_ctx_last_mm_used(al._h))
            bad = [i for i in range(batch.n) if got[i] != want[i]]
            only_mm = [i for i in bad if got[i][1] == want[i][1] and len(got[i][0]) == len(want[i][0]) and all(
                (h.tidx, h.toff, h.fw, h.cost, h.stratum, h.oms) == (y.tidx, y.toff, y.fw, y.cost, y.stratum, y.oms) for h, y in zip(got[i][0], want[i][0]))]
            rec.update(n_bad=len(bad), only_the_mismatch_list=len(only_mm), bad_among_second_pass_reads=sum(1 for i in bad if flagged[i]),
                       bad_among_first_pass_reads=sum(1 for i in bad if not flagged[i]), retried=int(al.last_retried))
            if bad:
                i = bad[0]
                rec["first"] = {"i": i, "second_pass": flagged[i], "got": repr(got[i])[:240], "want": repr(want[i])[:240]}
            n_bad_total += len(bad)
            cases.append(rec)
            torch.cuda.synchronize()
    out["cases"] = cases
    out["reads_that_differ"] = n_bad_total
    print(json.dumps(out))
    return 1 if n_bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
