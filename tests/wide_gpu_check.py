"""GPU checks of libbowtie_amd_l.so (64-bit BWT rows), run by tests/test_zz_wide_gpu.py in a process of its own with
BT_LIB=libbowtie_amd_l.so (the binding holds one library per process) and, for the biased runs, the wide loader's test knobs
in the environment (BT_WIDE_ROW_BIAS, BT_WIDE_SEG_SHIFT: tests/test_wide_rows_emu.py has the story).  Everything goes through
the C ABI; the oracle and the reference's golden outputs are the checkers.  usage: wide_gpu_check.py <what> [index]"""
import os
os.environ.setdefault("BT_TEST_KNOBS", "1")      # the loader honours BT_WIDE_ROW_BIAS / BT_WIDE_SEG_SHIFT only for tests
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import common as T                                            # noqa: E402
from bowtie_amd import _abi as A                              # noqa: E402
from bowtie_amd import aligner as AL                          # noqa: E402
from bowtie_amd.reads import Read, pack_reads                 # noqa: E402
from bowtie_amd.synth import synth_reads                      # noqa: E402


def phase_program(mode):
    kw = T.MODES[mode]
    return not (A.make_policy(**kw).best or kw.get("sample_max") or (kw.get("mode") == "v" and kw.get("mms") == 3))


def main():
    what = sys.argv[1]
    assert AL.lib().bt_rows64() == 1, "not the 64-bit-row library: BT_LIB=%r" % os.environ.get("BT_LIB")
    bias = int(os.environ.get("BT_WIDE_ROW_BIAS", "0"), 0)
    n_checked = 0
    if what == "rank":
        name = sys.argv[2]
        oi = T.oracle_index(name)
        ln = int(oi.fw.len)
        idx = AL.Index(os.path.join(T.G, name))
        assert AL.lib().bt_index_len64(idx._h) == ln
        al = AL.Aligner(idx, A.make_policy(**T.MODES["n2"]))
        rng = np.random.default_rng(5)
        rows = [int(r) for r in rng.integers(0, ln + 1, size=6000)] + [0, 1, 63, 64, 1023, 1024, int(oi.fw.zOff), int(oi.fw.zOff) + 1, ln]
        if bias:
            mid = (1 << 32) - bias
            rows += [mid - 1, mid, mid + 1, mid + 63, mid + 64]
        for mirror in (False, True):
            lf, L = al.probe_rank64(np.array([bias + r for r in rows], dtype=np.uint64), mirror)
            z = int(oi.ix(mirror).zOff)
            for i, r in enumerate(rows):
                olf, oL = oi.rank4(r, mirror)
                assert [int(v) for v in lf[i]] == [v + bias for v in olf], (name, mirror, r)
                if r != z:
                    assert int(L[i]) == oL, (name, mirror, r)
                n_checked += 1
    elif what == "golden":
        idx = {n: AL.Index(os.path.join(T.G, n)) for n in ("e_coli", "multi")}
        runs = [r for r in T.golden_runs(reads=("e_coli_1000", "syn100", "syn50lowq", "syn12", "syn150")) if phase_program(r["mode"])]
        if not os.environ.get("BT_WIDE_GPU_FULL"):
            runs = runs[::3]            # a third of them in the suite (every context of the wide build allocates twice the scratch): all of them on the host build
        for run in runs:
            batch = T.read_set(run["index"], run["reads"])
            kw = T.MODES[run["mode"]]
            res = AL.Aligner(idx[run["index"]], A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw))
            T.check_against_golden(run, res, batch, idx[run["index"]].refnames)
            n_checked += 1
    elif what == "family":
        import test_index_family as FAM
        idx = AL.Index(FAM.LARGE)
        for run in FAM.fam()["runs"]:
            if not phase_program(run["mode"]):
                continue
            batch = T.read_set("multi", run["reads"])
            kw = T.MODES[run["mode"]]
            res = AL.Aligner(idx, A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw))
            FAM._check(run, FAM._render(run, res, batch, idx.refnames))
            n_checked += 1
    elif what == "ragged":
        idx = AL.Index(os.path.join(T.G, "multi"))
        text = T.joined_text("multi")
        for mode in ("v0", "v1", "v2", "n2", "n3", "n2_k3", "n2_nomaq", "n1_a_m20"):
            kw = T.MODES[mode]
            rng = np.random.default_rng(99)
            reads = []
            for i in range(400):
                ln = int(rng.integers(4, 151))
                b = synth_reads(text, 1, ln, mm_dist=(0, 1, 2, 3), seed=1000 + i, n_frac=0.2, lowq_frac=0.1)
                reads.append(Read(("q%d" % i).encode(), b.seq[0, :ln].copy(), b.qual[0, :ln].tobytes()))
            # one batch of reads that fit the read-in-LDS builds, one that needs the register-window build
            for sub in ([r for r in reads if len(r.seq) <= 100], reads):
                batch = pack_reads(sub)
                c = A.OpCounts()
                import oracle_lib as OL
                oc = OL.OpCounts()
                got = AL.Aligner(idx, A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw), counts=c)
                want = T.oracle_results("multi", batch, kw, counts=oc)
                T.compare_results(got, want, "wide " + mode)
                for f in ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames"):
                    assert getattr(c, f) == getattr(oc, f), (mode, f, getattr(c, f), getattr(oc, f))
                assert c.loc_records == 0            # row space only
                n_checked += 1
    elif what == "second_pass":
        # one big batch, and the second pass for reads that outgrow their scratch (the streamed path with carry-over is the
        # bowtie-amd-l run of tests/test_zz_wide_gpu.py)
        idx = AL.Index(os.path.join(T.G, "e_coli"))
        text = T.joined_text("e_coli")
        kw = T.MODES["n2"]
        batch = synth_reads(text, 20000, 100, seed=77)
        want = T.oracle_results("e_coli", batch, kw)
        al = AL.Aligner(idx, A.make_policy(**kw))
        got = al.align(batch)
        T.compare_results(got, want, "wide n2 one batch")
        os.environ["BT_ENTRY_CAP"] = "12"
        al2 = AL.Aligner(idx, A.make_policy(**kw))
        got2 = al2.align(batch)
        del os.environ["BT_ENTRY_CAP"]
        T.compare_results(got2, want, "wide n2 with the second pass")
        assert AL.lib().bt_ctx_last_retried(al2._h) > 0
        n_checked += 2
    elif what == "best":
        # the best-first engine and pairs in the wide build: the .ebwtl goldens of bowtie-align-l (--best / --strata / -M / -v 3,
        # paired with --best), the plain index's paired goldens without --best (PairedBWAlignerV1)
        import test_index_family as FAM
        idx = AL.Index(FAM.LARGE)
        for run in FAM.fam()["runs"]:
            if phase_program(run["mode"]):
                continue
            batch = T.read_set("multi", run["reads"])
            kw = T.MODES[run["mode"]]
            res = AL.Aligner(idx, A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw))
            FAM._check(run, FAM._render(run, res, batch, idx.refnames))
            n_checked += 1
        for run in FAM.fam()["paired_runs"]:
            b1, b2 = T.pair_set("multi", run["reads"])
            kw = T.MODES[run["mode"]]
            res = AL.Aligner(idx, A.make_policy(**kw)).align_pairs(b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
            FAM._check(run, FAM._render_pairs(run, res, b1, b2, idx.refnames))
            n_checked += 1
        plain = AL.Index(os.path.join(T.G, "multi"))
        for run in T.paired_v1_runs("multi")[:6]:
            b1, b2 = T.pair_set("multi", run["reads"])
            kw = dict(T.MODES[run["mode"]], pe_v1=True)
            res = AL.Aligner(plain, A.make_policy(**kw)).align_pairs(b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
            T.check_pairs_against_golden(run, res, b1, b2, plain.refnames)
            n_checked += 1
    else:
        raise SystemExit("unknown check " + what)
    print("wide_gpu_check %s: ok, %d checks (row bias %d)" % (what, n_checked, bias))


if __name__ == "__main__":
    main()
