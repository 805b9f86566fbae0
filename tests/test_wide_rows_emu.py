"""64-bit BWT rows (the reference's bowtie-align-l, btypes.h:4-28), on the CPU: the host build of the per-read
automatons (phase programs, best-first engine, pairs) compiled with -DBT_WIDE=1 -- the sources libbowtie_amd_l.so is built
from -- against the oracle and against the unmodified reference's outputs.

No index of 2^32 rows fits a test suite, so the wide build's loader can number an image's rows from a bias (bt_host.h:
BT_WIDE_ROW_BIAS) and cut its rank blocks into small segments (BT_WIDE_SEG_SHIFT): a genome of a few Mbp then runs its whole
search -- ftab, LF, range stack, reported rows, the walk to a sampled row -- on row numbers on either side of 2^32, over
several segments, and must still give what the reference gives on the plain index (text offsets are not biased).  The run on
a real index of 4.5 x 10^9 rows is in profiles/r5/wide_rows_real_index.txt."""
import os

import numpy as np
import pytest

import common as T
import emu_lib as E
import oracle_lib as OL
import refrun as R
import test_index_family as FAM
from bowtie_amd import _abi as A
from bowtie_amd.reads import Read, pack_reads
from bowtie_amd.synth import synth_reads

SEG_SHIFT = 4            # 16 rank blocks = 1024 rows per segment


def _bias(name):
    """rows numbered so that 2^32 falls in the middle of the index"""
    half = int(T.oracle_index(name).fw.len) // 2
    g = 1 << (SEG_SHIFT + 6)
    return (1 << 32) - (half // g) * g


@pytest.fixture(scope="module")
def wide():
    out = {}
    for n in ("e_coli", "multi"):
        out[n, "plain"] = E.EmuAligner(T.G + "/" + n, wide=True)
        out[n, "biased"] = E.EmuAligner(T.G + "/" + n, wide=True, row_bias=_bias(n), seg_shift=SEG_SHIFT)
    return out


def test_the_wide_build_has_64_bit_rows(wide):
    for (n, kind), e in wide.items():
        ln, bias, width = e.dims()
        assert width == 8 and ln == T.oracle_index(n).fw.len
        assert bias == (_bias(n) if kind == "biased" else 0)
        if kind == "biased":
            assert bias < (1 << 32) < bias + ln


def test_wide_rank_vs_oracle(wide):
    """LF(row, ACGT) and the BWT character at every kind of row, rows below and above 2^32, every segment boundary near them"""
    rng = np.random.default_rng(7)
    for (name, kind), e in wide.items():
        oi = T.oracle_index(name)
        ln = int(oi.fw.len)
        bias = e.dims()[1]
        rows = list(rng.integers(0, ln + 1, size=2000)) + [0, 1, 63, 64, 1023, 1024, 1025, int(oi.fw.zOff), int(oi.fw.zOff) + 1, ln]
        if bias:
            mid = (1 << 32) - bias
            rows += [mid - 1, mid, mid + 1, mid + 63, mid + 64]
        for mirror in (False, True):
            z = int(oi.ix(mirror).zOff)
            for r in rows:
                lf, L = e.rank4(bias + int(r), mirror)
                olf, oL = oi.rank4(int(r), mirror)
                assert lf == [v + bias for v in olf], (name, kind, mirror, r)
                if r != z:
                    assert L == oL


RUNS = T.golden_runs(reads=("e_coli_1000", "syn100", "syn50lowq", "syn12", "syn150"))


def _phase_program(mode):
    """the modes the reference runs through its phase scripts (not the best-first workers: --best, -M, -v 3)"""
    kw = T.MODES[mode]
    return not (A.make_policy(**kw).best or kw.get("sample_max") or (kw.get("mode") == "v" and kw.get("mms") == 3))


@pytest.mark.parametrize("kind", ["plain", "biased"])
@pytest.mark.parametrize("run", [r for r in RUNS if _phase_program(r["mode"])], ids=lambda r: r["file"][:-7])
def test_wide_emu_matches_reference_sam(run, kind, wide):
    """the reference's own SAM for the plain index (bowtie-align-s: the files are 32-bit), rows 64 bits wide here"""
    batch = T.read_set(run["index"], run["reads"])
    kw = T.MODES[run["mode"]]
    res = wide[run["index"], kind].align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, n_lanes=37)
    T.check_against_golden(run, res, batch, T.oracle_index(run["index"]).refnames)


@pytest.mark.parametrize("kind", ["plain", "biased"])
@pytest.mark.parametrize("run", T.paired_runs(), ids=lambda r: r["file"][:-7])
def test_wide_emu_paired_matches_reference_sam(run, kind, wide):
    """paired-end with --best (PairedBWAlignerV2 + the window scan on the 2-bit reference) in the wide build"""
    b1, b2 = T.pair_set(run["index"], run["reads"])
    kw = T.MODES[run["mode"]]
    res = wide[run["index"], kind].align_pairs(A.make_policy(**kw), b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    T.check_pairs_against_golden(run, res, b1, b2, T.oracle_index(run["index"]).refnames)


@pytest.mark.parametrize("kind", ["plain", "biased"])
@pytest.mark.parametrize("run", T.paired_v1_runs(), ids=lambda r: r["file"][6:-7])
def test_wide_emu_paired_without_best_matches_reference_sam(run, kind, wide):
    """... and without --best (PairedBWAlignerV1, the reference's default paired aligner)"""
    b1, b2 = T.pair_set(run["index"], run["reads"])
    kw = dict(T.MODES[run["mode"]], pe_v1=True)
    res = wide[run["index"], kind].align_pairs(A.make_policy(**kw), b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    T.check_pairs_against_golden(run, res, b1, b2, T.oracle_index(run["index"]).refnames)


import test_automaton_emu as TA                         # noqa: E402


@pytest.mark.parametrize("mode", TA.BEST_RAGGED)
def test_wide_emu_best_first_vs_oracle_ragged(mode, wide):
    """ragged 1..150-base reads with Ns and low qualities through the wide best-first engine on biased rows: hits and op
    counts (same_pair apart: see below) equal the oracle's"""
    kw = T.MODES[mode]
    batch = TA.ragged_batch(300, 1, 151, 11)
    oc, ec = OL.OpCounts(), A.OpCounts()
    want = T.oracle_results("multi", batch, kw, cap=T.hit_cap_for(kw), counts=oc)
    got = wide["multi", "biased"].align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), counts=ec)
    T.compare_results(got, want, mode)
    for f in ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames"):
        assert getattr(ec, f) == getattr(oc, f), (mode, f, getattr(ec, f), getattr(oc, f))


@pytest.mark.parametrize("mode", ["pe_n1_best_X500", "pe_n2_best_X400_I250_k3", "pe_v3_best_X500", "pe_n1_a_strata_X500", "pev1_n2_X500", "pev1_n1_X500_a"])
def test_wide_emu_paired_vs_oracle_counts(mode, wide):
    kw = T.MODES[mode]
    v1 = mode.startswith("pev1")
    b1, b2 = T.pair_set("multi", "pe50")
    oc, ec = OL.OpCounts(), A.OpCounts()
    cap = 2048 if kw.get("all_hits") else None
    want = T.oracle_pair_results("multi", b1, b2, kw, cap=cap, counts=oc, v1=v1)
    got = wide["multi", "biased"].align_pairs(A.make_policy(**(dict(kw, pe_v1=True) if v1 else kw)), b1, b2, hit_cap=cap, counts=ec)
    T.compare_results(got, want, mode)
    for f in ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames"):
        assert getattr(ec, f) == getattr(oc, f), (mode, f, getattr(ec, f), getattr(oc, f))


@pytest.mark.parametrize("no_rl", [False, True], ids=["read_in_lds", "register_window"])
@pytest.mark.parametrize("mode", ["v0", "v1", "v2", "n2", "n3", "n2_k3", "n2_nomaq", "n1_a_m20"])
def test_wide_emu_vs_oracle_ragged(mode, no_rl, wide):
    """Ragged lengths (4..150), Ns, low qualities on biased rows; results and op counts equal the oracle's (same_pair apart:
    it asks whether two rows share a 448-row side pair, which a bias that is no multiple of 448 moves)"""
    kw = T.MODES[mode]
    text = T.joined_text("multi")
    rng = np.random.default_rng(99)
    reads = []
    for i in range(300):
        L = int(rng.integers(4, 151 if no_rl else 113))
        b = synth_reads(text, 1, L, mm_dist=(0, 1, 2, 3), seed=1000 + i, n_frac=0.2, lowq_frac=0.1)
        reads.append(Read(("q%d" % i).encode(), b.seq[0, :L].copy(), b.qual[0, :L].tobytes()))
    batch = pack_reads(reads)
    pol = A.make_policy(**kw)
    counts = A.OpCounts()
    got = wide["multi", "biased"].align(pol, batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, counts=counts, no_rl=no_rl)
    oc = OL.OpCounts()
    want = T.oracle_results("multi", batch, kw, counts=oc)
    T.compare_results(got, want, "wide " + mode)
    for f in ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames"):
        assert getattr(counts, f) == getattr(oc, f), (mode, f, getattr(counts, f), getattr(oc, f))


@pytest.mark.parametrize("bias", [False, True], ids=["plain", "biased"])
@pytest.mark.parametrize("run", [r for r in FAM.fam()["runs"] if r["reads"] in ("syn36", "syn150", "syn50lowq")],
                         ids=lambda r: os.path.basename(r["file"])[:-7])
def test_wide_emu_on_large_index_matches_bowtie_align_l(run, bias):
    """a .ebwtl index (64-bit offsets on disk, the reference's bowtie-build-l) against bowtie-align-l's own output: the
    two-draw row choice of the 64-bit binary included"""
    ln = int(OL.OracleIndex(os.path.join(T.G, "multi"), wide=True).fw.len)
    g = 1 << (SEG_SHIFT + 6)
    emu = E.EmuAligner(FAM.LARGE, wide=True, row_bias=((1 << 32) - (ln // 2 // g) * g) if bias else None, seg_shift=SEG_SHIFT if bias else None)
    batch = T.read_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = emu.align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, n_lanes=37)
    FAM._check(run, FAM._render(run, res, batch, FAM.wide_oracle().refnames))


@pytest.mark.parametrize("kind", ["plain", "biased"])
@pytest.mark.parametrize("run", [r for r in RUNS if not _phase_program(r["mode"])], ids=lambda r: r["file"][:-7])
def test_wide_emu_best_first_matches_reference_sam(run, kind, wide):
    """the best-first engine (--best, --strata, -M, -v 3) with 64-bit rows: branch records, alternatives and the chaser hold
    rows as two words"""
    batch = T.read_set(run["index"], run["reads"])
    kw = T.MODES[run["mode"]]
    res = wide[run["index"], kind].align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw))
    T.check_against_golden(run, res, batch, T.oracle_index(run["index"]).refnames)


@pytest.mark.parametrize("bias", [False, True], ids=["plain", "biased"])
@pytest.mark.parametrize("run", FAM.fam()["paired_runs"], ids=lambda r: os.path.basename(r["file"])[:-7])
def test_wide_emu_paired_on_large_index_matches_bowtie_align_l(run, bias):
    ln = int(OL.OracleIndex(os.path.join(T.G, "multi"), wide=True).fw.len)
    g = 1 << (SEG_SHIFT + 6)
    emu = E.EmuAligner(FAM.LARGE, wide=True, row_bias=((1 << 32) - (ln // 2 // g) * g) if bias else None, seg_shift=SEG_SHIFT if bias else None)
    b1, b2 = T.pair_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = emu.align_pairs(A.make_policy(**kw), b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    FAM._check(run, FAM._render_pairs(run, res, b1, b2, FAM.wide_oracle().refnames))


def test_wide_loader_refuses_a_bias_that_does_not_fit_the_segments():
    with pytest.raises(IOError):
        E.EmuAligner(T.G + "/multi", wide=True, row_bias=(1 << 32) + 64, seg_shift=SEG_SHIFT)


def test_the_wide_binary_through_the_cpu_shim():
    """bowtie-amd-l (tests/test_zz_wide_gpu.py's binary test) without a GPU: its host logic over the wide emulator's searches,
    LD_PRELOADed as tests/test_cli_shim.py does for bowtie-amd"""
    import subprocess
    import sys
    env = dict(os.environ, BT_TEST_CLI_SHIM="1", LD_PRELOAD=E.wide_shim())
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-n", "0", "tests/test_zz_wide_gpu.py", "-k", "cli_l"],
                       cwd=T.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    tail = p.stdout.decode(errors="replace")[-2500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail


# ---- differential fuzz against the live 64-bit reference: bowtie-build-l + bowtie-align-l ------------------------------------
BUILD_L = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-build-l")
ALIGN_L = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-align-l")


@pytest.mark.skipif(not os.path.exists(ALIGN_L), reason="needs the reference binaries (make -C oracle ref)")
@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_FUZZ_OFFSET", "0")), int(os.environ.get("BT_FUZZ_OFFSET", "0")) + int(os.environ.get("BT_WIDE_FUZZ_SEEDS", "60"))))
def test_wide_engine_against_bowtie_align_l(seed, tmp_path):
    """tests/test_engine_fuzz.py's small random genomes (several sequences, N gaps, repeats, lengths from 8 bases up: the '$'
    row, the eftab, fragment ends, reads longer than the genome), indexed by the reference's bowtie-build-l with random
    ftab / SA-sample rates, searched by bowtie-align-l with a random phase-program policy, report mode and output options --
    against the wide host build on the same .ebwtl files with its rows numbered from a random bias over random segment
    sizes, through the product's parser and formatter."""
    import random
    import subprocess
    import cli_cases as CC
    import test_engine_fuzz as F
    from bowtie_amd import hostio as H
    rng = random.Random(9000 + seed)
    seqs = F.make_genome(rng)
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%s\n%s\n" % ("s%d some description" % i if i % 2 == 0 else "t%d" % i, s))
    base = str(tmp_path / "g")
    ftab, off = rng.choice([1, 2, 3, 4, 6]), rng.choice([1, 2, 3, 5])
    b = subprocess.run([BUILD_L, "--ftabchars", str(ftab), "--offrate", str(off), "-q", fa, base], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if b.returncode != 0:
        pytest.skip("bowtie-build-l refuses this genome: " + b.stderr.decode(errors="replace")[-200:])
    lens = [4, 5, 7, 10, 12, 16, 22, 30]
    if max(len(g) for g in seqs) >= 150 and rng.random() < 0.5:
        lens = [30, 60, 105, 110, 113, 130]
    reads = F.make_reads(rng, seqs, rng.randrange(4, 14), lens)
    fq = str(tmp_path / "r.fq")
    F._write_fastq(fq, reads)
    seg_shift = rng.choice([2, 3, 5])
    gran = 1 << max(seg_shift + 6, off + 1)
    bias = rng.choice([0, (1 << 32) - gran * rng.randrange(0, 3), (1 << 33) + gran * rng.randrange(0, 5), (1 << 36) - gran])
    emu = E.EmuAligner(base, wide=True, row_bias=bias, seg_shift=seg_shift)
    done = 0
    for _ in range(4):
        pol_args = rng.choice(F.UNPAIRED_POLICIES)                      # both engines: the phase scripts and --best / --strata / -M / -v 3
        rep = [x for x in rng.choice(F.REPORTS)]
        if "-M" in pol_args or "-m" in pol_args or ("-k" in pol_args and "-k" in rep):
            rep = [x for x in rep if x not in ("-m", "-k", "1", "2", "3")] if ("-M" in pol_args or "-m" in pol_args) else []
        args = pol_args + rep + F.out_options(rng) + ["--seed", str(rng.randrange(0, 5))]
        if not F._args_ok(args):
            continue
        ref = subprocess.run([ALIGN_L, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if ref.returncode != 0:
            assert ref.returncode > 0 and (b"is less than" in ref.stderr or b"at least" in ref.stderr), (args, ref.stderr[-300:])
            continue
        rd, pol, out, ex = CC.interpret(args)
        b1 = H.read_all(fq, **rd)
        oi = type("Refs", (), {})()
        oi.refnames, oi.reflens = emu.refs()
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        p = A.make_policy(**pol)
        per = emu.align(p, b1, hit_cap=cap, lite=(not p.best and rng.random() < 0.5), no_rl=(not p.best and rng.random() < 0.3))
        hits, nh, st, pool = H.pack_hits(per, cap)
        got, tally = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
        got = F._header_for(ref.stdout, oi, opts, ex) + got
        assert got == ref.stdout, (seqs, args, bias, seg_shift)
        assert H.summary(tally).strip().split("\n") == F._summary_of(ref.stderr), args
        done += 1
    assert done or True


@pytest.mark.skipif(not os.path.exists(ALIGN_L), reason="needs the reference binaries (make -C oracle ref)")
@pytest.mark.parametrize("best", [True, False], ids=["best", "v1"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_FUZZ_OFFSET", "0")), int(os.environ.get("BT_FUZZ_OFFSET", "0")) + int(os.environ.get("BT_WIDE_FUZZ_SEEDS", "60")) // 2))
def test_wide_paired_engine_against_bowtie_align_l(seed, best, tmp_path):
    """tests/test_engine_fuzz.py's paired fuzz (PairedBWAlignerV2 with --best, PairedBWAlignerV1 without), the index by
    bowtie-build-l (with its .3/.4.ebwtl), the answers by bowtie-align-l, the wide host build on biased rows"""
    import random
    import subprocess
    import cli_cases as CC
    import test_engine_fuzz as F
    from bowtie_amd import hostio as H
    rng = random.Random((30_000 if best else 40_000) + seed)
    seqs = [s for s in F.make_genome(rng)]
    seqs.append("".join(rng.choice("ACGT") for _ in range(rng.choice([60, 120, 250]))))
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        for i, sq in enumerate(seqs):
            f.write(">%s\n%s\n" % ("s%d some description" % i if i % 2 == 0 else "t%d" % i, sq))
    base = str(tmp_path / "g")
    off = rng.choice([1, 3, 5])
    b = subprocess.run([BUILD_L, "--ftabchars", str(rng.choice([1, 2, 4, 6])), "--offrate", str(off), "-q", fa, base], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if b.returncode != 0:
        pytest.skip("bowtie-build-l refuses this genome")
    m1, m2 = [], []
    for i in range(rng.randrange(3, 10)):
        g = rng.choice(seqs)
        L1, L2 = rng.choice([5, 8, 12, 20]), rng.choice([5, 8, 12, 20])
        Fr = rng.randrange(max(L1, L2), max(L1, L2) + 60)
        if len(g) >= Fr and rng.random() < 0.85:
            p0 = rng.randrange(0, len(g) - Fr + 1)
            frag = g[p0:p0 + Fr].replace("N", "C")
            a, bb = list(frag[:L1]), list(F._rc(frag[Fr - L2:]))
            for sq in (a, bb):
                for _ in range(rng.choice([0, 0, 1, 2])):
                    sq[rng.randrange(len(sq))] = rng.choice("ACGT")
            a, bb = "".join(a), "".join(bb)
            if rng.random() < 0.3:
                a, bb = bb, a
        else:
            a = "".join(rng.choice("ACGT") for _ in range(L1)); bb = "".join(rng.choice("ACGT") for _ in range(L2))
        m1.append(("p%d" % i, a, "".join(rng.choice("!+5?IIII") for _ in a)))
        m2.append(("p%d" % i, bb, "".join(rng.choice("!+5?IIII") for _ in bb)))
    f1, f2 = str(tmp_path / "m_1.fq"), str(tmp_path / "m_2.fq")
    F._write_fastq(f1, m1, 1)
    F._write_fastq(f2, m2, 2)
    seg_shift = rng.choice([2, 3, 5])
    gran = 1 << max(seg_shift + 6, off + 1)
    bias = rng.choice([0, (1 << 32) - gran * rng.randrange(0, 3), (1 << 33) + gran * rng.randrange(0, 5), (1 << 36) - gran])
    emu = E.EmuAligner(base, wide=True, row_bias=bias, seg_shift=seg_shift)
    refs = type("Refs", (), {})()
    refs.refnames, refs.reflens = emu.refs()
    for _ in range(3):
        args = rng.choice(F.PAIRED_POLICIES) + (["--best"] if best else []) + rng.choice(F.PAIRED_REPORTS) + \
            rng.choice([["-X", "100"], ["-X", "60", "-I", "10"], ["-X", "250"]]) + rng.choice([[], [], ["-5", "1"], ["-3", "2"], ["-5", "2", "-3", "1"]]) + F.out_options(rng)
        if not F._args_ok(args):
            continue
        if not best and "--strata" in args and "-M" not in args and args[:2] != ["-v", "3"]:
            continue
        ref = subprocess.run([ALIGN_L, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, "-1", f1, "-2", f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if ref.returncode != 0:
            assert ref.returncode > 0 and (b"is less than" in ref.stderr or b"at least" in ref.stderr), (args, ref.stderr[-300:])
            continue
        rd, pol, out, ex = CC.interpret(args)
        b1, b2 = H.read_all(f1, mate=1, **rd), H.read_all(f2, mate=2, **rd)
        o1 = rd.get("trim5", 0) if pol.get("mate1_fw", True) else rd.get("trim3", 0)
        o2 = rd.get("trim3", 0) if pol.get("mate2_fw", False) else rd.get("trim5", 0)
        pol = dict(pol, min_ins=max(0, max(0, pol.get("min_ins", 0) - o1) - o2), max_ins=max(0, max(0, pol.get("max_ins", 250) - o1) - o2))
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else 2 * max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        if not best:
            pol = dict(pol, pe_v1=True)
        per = emu.align_pairs(A.make_policy(**pol), b1, b2, hit_cap=cap)
        hits, nh, st, pool = H.pack_hits(per, cap)
        got, tally = H.format_pairs(b1, b2, hits, nh, st, pool, cap, refs.refnames, refs.reflens, opts)
        got = F._header_for(ref.stdout, refs, opts, ex) + got
        assert got == ref.stdout, (seqs, args, bias, seg_shift)
        assert H.summary(tally).strip().split("\n") == F._summary_of(ref.stderr), args


@pytest.mark.skipif(not TA._msan_available(), reason="needs clang with the MemorySanitizer runtime")
def test_wide_automaton_reads_nothing_uninitialised(tmp_path):
    """tests/emu/emu_msan.cpp with -DBT_WIDE=1: LDS, the scratch arenas and what the kernel's rank branch leaves unset of a
    round's answer (in the wide build: the second row's two pieces) poisoned; 40 reads with Ns on biased rows, three
    policies, all three builds of the automaton"""
    import subprocess
    exe = str(tmp_path / "emu_msan_w")
    subprocess.check_call([TA.MSAN_CLANG, "-fsanitize=memory", "-fsanitize-recover=memory", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17", "-w",
                           "-DBT_WIDE=1", "-pthread", "-o", exe, os.path.join(T.ROOT, "tests", "emu", "emu_msan.cpp"), os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_host.cpp")])
    b = synth_reads(T.joined_text("multi"), 40, 60, mm_dist=(0, 1, 2, 3), seed=5, n_frac=0.1)
    reads = ["".join("ACGTN"[c] for c in b.seq[i, :60]) for i in range(b.n)]
    env = dict(os.environ, MSAN_OPTIONS="halt_on_error=0:exitcode=0", BT_WIDE_ROW_BIAS=str(_bias("multi")), BT_WIDE_SEG_SHIFT=str(SEG_SHIFT), BT_LOAD_THREADS="1")
    for pol in (["n", "2", "0", "1"], ["v", "2", "1", "1"], ["n", "3", "0", "3"]):
        for rl_mode in ("2", "0", "1"):
            p = subprocess.run([exe, os.path.join(T.G, "multi")] + pol + ["64", rl_mode] + reads, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            frames0 = [ln for ln in p.stderr.decode(errors="replace").splitlines() if ln.lstrip().startswith("#0 ")]
            mine = [ln for ln in frames0 if "File::File" not in ln and "operator new" not in ln]
            assert not mine, "\n".join(mine[:5])
            assert "rc 0" in p.stdout.decode()


@pytest.mark.parametrize("what", [["best"], ["family"], ["rank", "multi"], ["ragged"]], ids=lambda w: w[0])
def test_the_gpu_check_script_on_the_host_build(what):
    """tests/wide_gpu_check.py -- what tests/test_zz_wide_gpu.py runs against libbowtie_amd_l.so -- with the binding stubbed by
    the wide host build: the script's own logic, where there is no GPU"""
    import subprocess
    import sys
    env = dict(os.environ, BT_WIDE_ROW_BIAS=str(_bias("multi")), BT_WIDE_SEG_SHIFT=str(SEG_SHIFT))
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "tests", "wide_gpu_check_dry.py")] + what, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b": ok, " in p.stdout, p.stdout.decode()[-3000:]


def _fake_huge_index(base, rows=(1 << 32) + 12345):
    """the first bytes of a .ebwtl index that claims `rows` BWT rows: as far as a loader reads before it knows the size"""
    import struct
    for ext in ("", ".rev"):
        with open(base + ext + ".1.ebwtl", "wb") as f:
            f.write(struct.pack("<iQiiiii", 1, rows - 1, 7, 1, 5, 10, 0))
        with open(base + ext + ".2.ebwtl", "wb") as f:
            f.write(struct.pack("<i", 1))


def test_the_32_bit_library_points_an_index_of_2_to_32_rows_to_the_64_bit_one(tmp_path):
    """BT_ERR_ROWS64 from the 32-bit loader (host-only entry point: no GPU needed); the 64-bit library reads on (and finds
    this file truncated)"""
    import ctypes as C
    from bowtie_amd import aligner as AL
    base = str(tmp_path / "huge")
    _fake_huge_index(base)
    out = (C.c_uint64 * 8)()
    assert AL.lib().bt_index_digest(base.encode(), 0, out) == A.BT_ERR_ROWS64
    wide = C.CDLL(os.path.join(os.path.dirname(AL.LIB_PATH), "libbowtie_amd_l.so"))
    wide.bt_index_digest.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
    assert wide.bt_index_digest(base.encode(), 0, out) in (A.BT_ERR_IO, A.BT_ERR_FORMAT)
    # one row below the limit is the 32-bit build's own (and truncated here)
    _fake_huge_index(base, rows=(1 << 32) - 1)
    assert AL.lib().bt_index_digest(base.encode(), 0, out) in (A.BT_ERR_IO, A.BT_ERR_FORMAT)


def test_bowtie_amd_starts_bowtie_amd_l_for_such_an_index(tmp_path):
    """the reference's wrapper picks bowtie-align-l (bowtie:52-81); bowtie-amd execs bowtie-amd-l.  Under the 32-bit CPU shim both
    binaries get BT_ERR_ROWS64 from the loader: the first starts the second, the second (whose library has 64-bit rows)
    reports it -- seen from outside as the message of the second"""
    import subprocess
    base = str(tmp_path / "huge")
    _fake_huge_index(base)
    fq = str(tmp_path / "r.fq")
    with open(fq, "w") as f:
        f.write("@r0\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n")
    env = dict(os.environ, LD_PRELOAD=E.shim())
    p = subprocess.run([os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd"), "-x", base, fq], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0 and "could not be started" not in err and "2^32-1 rows" in err, err


def test_bowtie_amd_picks_the_binary_before_it_takes_a_read_from_a_pipe(tmp_path):
    """Reads on standard input, from a pipe whose writer has sent nothing yet: bowtie-amd must start bowtie-amd-l for an index
    of 2^32 rows WITHOUT waiting for -- let alone consuming -- a first batch (what it has read from a pipe the second binary
    never sees; the reference's wrapper decides from the file names before anything runs, bowtie:52-81).  Until round 6 the
    first batch was parsed before the decision: this call then blocks on the silent pipe until the timeout."""
    import subprocess
    base = str(tmp_path / "huge")
    _fake_huge_index(base)
    env = dict(os.environ, LD_PRELOAD=E.shim())
    p = subprocess.Popen([os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd"), "-x", base, "-"], env=env, stdin=subprocess.PIPE,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        p.wait(timeout=60)                       # nothing is written to its stdin, which stays open
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
        raise AssertionError("bowtie-amd waited for reads from the pipe before it chose the 64-bit-row binary")
    err = p.stderr.read().decode(errors="replace")
    p.stdin.close()
    assert p.returncode != 0 and "could not be started" not in err and "2^32-1 rows" in err, err


@pytest.mark.parametrize("wide_build", [False, True], ids=["rows32", "rows64"])
def test_carry_over_round_trip_on_the_host_build(wide_build):
    """EMU_PARK_EVERY (tests/emu/bt_emu.cpp): every lane is parked and adopted again in one round out of three, at random -- its
    state through the pool record's bytes as the kernel copies it (only what lies before the register window in the builds that
    keep the read in LDS), LDS lost, the read loaded again -- in both widths of the row type; results are the oracle's"""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import common as T, emu_lib as E
from bowtie_amd import _abi as A
wide = %r
e = E.EmuAligner(T.G + "/multi", wide=wide, row_bias=((1 << 32) - 29696) if wide else None, seg_shift=4 if wide else None)
n = 0
for mode in ("n2", "v2", "n3", "n2_k3", "n1_a_m20"):
    kw = T.MODES[mode]
    for reads, kwargs in (("syn100", {}), ("syn100", {"lite": True}), ("syn150", {"no_rl": True})):
        batch = T.read_set("multi", reads)
        got = e.align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, n_lanes=37, **kwargs)
        T.compare_results(got, T.oracle_results("multi", batch, kw, cap=T.hit_cap_for(kw)), mode)
        n += 1
print("ok", n)
''' % (T.ROOT, os.path.join(T.ROOT, "tests"), wide_build)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EMU_PARK_EVERY="3"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"ok 15" in p.stdout, p.stdout.decode()[-3000:]


def test_large_index_option_takes_the_ebwtl_files(tmp_path):
    """bowtie --large-index (the wrapper then runs bowtie-align-l, bowtie:64-65): with both builds of an index under one base,
    bowtie-amd answers as bowtie-align-s does on the .ebwt files, bowtie-amd --large-index starts bowtie-amd-l, whose loader
    takes the .ebwtl files and answers as bowtie-align-l does (the two-draw choice of the reported row tells them apart).
    Through the CPU shims."""
    import hashlib
    import shutil
    import subprocess
    from bowtie_amd.synth import write_fastq
    base = str(tmp_path / "both")
    for ext in ("1", "2", "3", "4", "rev.1", "rev.2"):
        shutil.copy(os.path.join(T.G, "multi." + ext + ".ebwt"), base + "." + ext + ".ebwt")
        shutil.copy(FAM.LARGE + "." + ext + ".ebwtl", base + "." + ext + ".ebwtl")
    small = {(r["reads"], r["mode"]): r for r in T.golden_runs("multi")}
    picked = [r for r in FAM.fam()["runs"] if not r["same_as_small"] and (r["reads"], r["mode"]) in small and _phase_program(r["mode"])][:2]
    assert picked
    fq = str(tmp_path / "r.fq")
    binp = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")
    for run in picked:
        write_fastq(T.read_set("multi", run["reads"]), fq)
        outs = {}
        for tag, extra, shim in (("small", [], E.shim()), ("large", ["--large-index"], E.wide_shim())):
            # (the exec'd bowtie-amd-l keeps the environment: it gets the wide shim from the start)
            p = subprocess.run([binp, "-S", "--sam-nohead"] + extra + run["args"] + ["-x", base, fq], env=dict(os.environ, LD_PRELOAD=shim),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert p.returncode == 0, p.stderr.decode()
            outs[tag] = hashlib.md5(p.stdout).hexdigest()
        assert outs["large"] == run["md5"], run["file"]
        assert outs["small"] != outs["large"]
    # a base with no 64-bit files: bowtie-align-l -- what --large-index makes the reference's wrapper run -- does not find it
    only_small = str(tmp_path / "small")
    for ext in ("1", "2", "3", "4", "rev.1", "rev.2"):
        shutil.copy(os.path.join(T.G, "multi." + ext + ".ebwt"), only_small + "." + ext + ".ebwt")
    p = subprocess.run([binp, "-S", "--large-index", "-x", only_small, fq], env=dict(os.environ, LD_PRELOAD=E.wide_shim()),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"Could not locate a Bowtie index" in p.stderr, p.stderr.decode()


def test_wide_loader_survives_damaged_index_files(tmp_path):
    """tests/emu/index_asan.cpp with -DBT_WIDE=1: the wide loader (64-bit tables, rank blocks and segment table built in threads
    from the file's BWT, the row bias) under AddressSanitizer + UBSan on 250 damaged copies of a bowtie-build-l index -- it may
    load them or refuse them, not crash"""
    import random
    import subprocess
    if not os.path.exists(BUILD_L):
        pytest.skip("needs oracle/_ref/bowtie-build-l")
    exe = str(tmp_path / "index_asan_w")
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17", "-w", "-pthread", "-DBT_WIDE=1",
                        "-o", exe, os.path.join(T.ROOT, "tests", "emu", "index_asan.cpp"), os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_host.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        f.write(">a x\nAGCATCGATCAGTATCTGACCNNNGTTAGGCATTACGGATCCATGCAAGTCTTGACGTACGGTCAATGC\n>b\nACGTTGCAAC\n")
    subprocess.run([BUILD_L, "--ftabchars", "3", "--offrate", "2", "-q", fa, str(tmp_path / "g")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    exts = ["1.ebwtl", "2.ebwtl", "3.ebwtl", "4.ebwtl", "rev.1.ebwtl", "rev.2.ebwtl"]
    orig = {e: open(str(tmp_path / "g") + "." + e, "rb").read() for e in exts}
    w = str(tmp_path / "w")
    seen = set()
    for seed in range(250):
        rng = random.Random(seed)
        for e in exts:
            with open(w + "." + e, "wb") as f:
                f.write(orig[e])
        e = rng.choice(exts)
        b = bytearray(orig[e])
        kind = rng.choice(["flip", "trunc", "word", "zero", "grow"])
        if kind == "flip":
            for _ in range(rng.choice([1, 1, 2, 5])):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == "trunc":
            b = b[:rng.randrange(0, len(b) + 1)]
        elif kind == "word" and len(b) >= 8:
            i = rng.randrange(0, min(len(b) - 7, 120))
            b[i:i + 8] = rng.choice([bytes([255]) * 8, bytes(8), (1 << 40).to_bytes(8, "little"), (0x7fffffff).to_bytes(8, "little")])
        elif kind == "zero":
            b = bytearray(len(b))
        else:
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 50)))
        with open(w + "." + e, "wb") as f:
            f.write(b)
        env = dict(os.environ, BT_LOAD_THREADS=str(rng.choice([1, 3])))
        if rng.random() < 0.3:
            env.update(BT_WIDE_SEG_SHIFT="2", BT_WIDE_ROW_BIAS=str(1 << 32))
        p = subprocess.run([exe, w], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60, env=env)
        assert p.returncode == 0 and b"ERROR" not in p.stderr and b"runtime error" not in p.stderr, (seed, e, kind, p.stderr.decode(errors="replace")[-800:])
        seen.add(p.stdout.split(b" len")[0])
    assert len(seen) >= 3
