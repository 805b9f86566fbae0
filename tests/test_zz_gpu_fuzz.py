"""GPU: the bowtie-amd binary against the unmodified reference binary, live, on seeded random tiny genomes (the generators
of test_engine_fuzz.py: '$' row, eftab, fragment ends, reads longer than the genome all come up constantly) with random
reads, pairs and option sets -- stdout and the summary on stderr, byte for byte.  Unpaired through both engines (a third of
the default-engine runs through --stream, i.e. the carry-over kernel instances), pairs through --best.  oracle/_ref
travels to the GPU box with the repository.  Named to run last: what it could find is rare-path trouble in the kernels."""
import os
import random
import subprocess

import numpy as np
import pytest

import common as T
import test_engine_fuzz as F
from bowtie_amd import ebwt_build as EB

BIN = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(F.REF_BIN), reason="needs oracle/_ref/bowtie-align-s")]


def _both(args, tail, extra=()):
    ref = subprocess.run([F.REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    got = subprocess.run([BIN, "--wrapper", "basic-0", "-p", "1"] + list(extra) + args + tail, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return ref, got


def _check(ref, got, what):
    if ref.returncode != 0:
        # a read shorter than the mode allows: both stop with the reference's message
        assert ref.returncode > 0 and got.returncode == 1, (what, ref.stderr[-200:], got.stderr[-200:])
        return
    assert got.returncode == 0, (what, got.stderr.decode(errors="replace")[-400:])
    strip_pg = lambda b: b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))     # @PG quotes the command line
    assert strip_pg(got.stdout) == strip_pg(ref.stdout), what
    assert F._summary_of(got.stderr) == F._summary_of(ref.stderr), what


@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_GPU_FUZZ_SEEDS", "60"))))
def test_binary_against_the_reference_unpaired(seed, tmp_path):
    rng = random.Random(seed)
    seqs = F.make_genome(rng)
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs], ["s%d some description" % i if i % 2 == 0 else "t%d" % i for i in range(len(seqs))],
                   base, ftab_chars=rng.choice([1, 2, 3, 4, 6]), off_rate=rng.choice([1, 2, 3, 5]))
    lens = [4, 5, 7, 10, 12, 16, 22, 30]
    if max(len(g) for g in seqs) >= 150 and rng.random() < 0.5:
        lens = [30, 60, 105, 110, 113, 130]
    reads = F.make_reads(rng, seqs, rng.randrange(4, 14), lens)
    fq = str(tmp_path / "r.fq")
    F._write_fastq(fq, reads)
    for _ in range(3):
        pol_args = rng.choice(F.UNPAIRED_POLICIES)
        rep = [x for x in rng.choice(F.REPORTS)]
        if "-M" in pol_args or "-m" in pol_args or ("-k" in pol_args and "-k" in rep):
            rep = [x for x in rep if x not in ("-m", "-k", "1", "2", "3")] if ("-M" in pol_args or "-m" in pol_args) else []
        args = pol_args + rep + F.out_options(rng) + ["--seed", str(rng.randrange(0, 5))]
        if not F._args_ok(args):
            continue
        stateful = "--best" in args or "--strata" in args or "-M" in args or args[:2] == ["-v", "3"]
        extra = ["--no-stream"] if (not stateful and rng.random() < 0.34) else []      # streaming is the default
        extra = extra + rng.choice([[], [], ["--batch", "3"], ["--batch", "2", "--inflight", "3"]])      # tiny batches: boundaries, order
        ref, got = _both(args, ["-x", base, fq], extra)
        _check(ref, got, (seqs, extra + args))


@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_GPU_FUZZ_SEEDS", "40"))))
def test_binary_against_the_reference_paired(seed, tmp_path):
    _paired(seed, tmp_path, True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_GPU_FUZZ_SEEDS", "40"))))
def test_binary_against_the_reference_paired_without_best(seed, tmp_path):
    """The reference's default paired-end aligner, PairedBWAlignerV1."""
    _paired(seed, tmp_path, False)


def _paired(seed, tmp_path, best):
    rng = random.Random((10_000 if best else 20_000) + seed)
    seqs = [s for s in F.make_genome(rng)]
    seqs.append("".join(rng.choice("ACGT") for _ in range(rng.choice([60, 120, 250]))))
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs], ["s%d" % i for i in range(len(seqs))], base,
                   ftab_chars=rng.choice([1, 2, 4, 6]), off_rate=rng.choice([1, 3, 5]))
    m1, m2 = [], []
    for i in range(rng.randrange(3, 10)):
        g = rng.choice(seqs)
        L1, L2 = rng.choice([5, 8, 12, 20]), rng.choice([5, 8, 12, 20])
        Fr = rng.randrange(max(L1, L2), max(L1, L2) + 60)
        if len(g) >= Fr and rng.random() < 0.85:
            p = rng.randrange(0, len(g) - Fr + 1)
            frag = g[p:p + Fr].replace("N", "C")
            a, b = list(frag[:L1]), list(F._rc(frag[Fr - L2:]))
            for s in (a, b):
                for _ in range(rng.choice([0, 0, 1, 2])):
                    s[rng.randrange(len(s))] = rng.choice("ACGT")
            a, b = "".join(a), "".join(b)
        else:
            a = "".join(rng.choice("ACGT") for _ in range(L1)); b = "".join(rng.choice("ACGT") for _ in range(L2))
        m1.append(("p%d" % i, a, "".join(rng.choice("!+5?IIII") for _ in a)))
        m2.append(("p%d" % i, b, "".join(rng.choice("!+5?IIII") for _ in b)))
    f1, f2 = str(tmp_path / "m_1.fq"), str(tmp_path / "m_2.fq")
    F._write_fastq(f1, m1, 1)
    F._write_fastq(f2, m2, 2)
    for _ in range(3):
        args = rng.choice(F.PAIRED_POLICIES) + (["--best"] if best else []) + rng.choice(F.PAIRED_REPORTS) + \
            rng.choice([["-X", "100"], ["-X", "60", "-I", "10"], ["-X", "250"]]) + rng.choice([[], [], ["-5", "1"], ["-3", "2"]]) + F.out_options(rng)
        if not F._args_ok(args):
            continue
        if not best and "--strata" in args and "-M" not in args and args[:2] != ["-v", "3"]:
            continue            # "--strata must be combined with --best" unless -v 3 / -M made the run stateful already
        ref, got = _both(args, ["-x", base, "-1", f1, "-2", f2], rng.choice([[], [], ["--batch", "2"], ["--batch", "3", "--inflight", "1"]]))
        _check(ref, got, (seqs, args))


@pytest.mark.parametrize("best", [True, False], ids=["best", "without_best"])
def test_binary_pairs_with_more_alignments_than_the_uniform_slots(best, tmp_path):
    """A tandem repeat gives every pair dozens of paired alignments: under -a / a large -k the pairs that outgrow the
    batch's uniform hit slots are searched again on their own with room for all (the whole batch used to be widened,
    and a 4 M-pair batch then refused with "lower --batch")."""
    rng = random.Random(99)
    unit = "".join(rng.choice("ACGT") for _ in range(37))
    flank = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    g = flank(60) + unit * 14 + flank(60)
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(g.encode(), dtype=np.uint8)]], ["rep"], base, ftab_chars=4, off_rate=2)
    m1, m2 = [], []
    for i in range(6):
        p = 60 + 37 * rng.randrange(0, 3) + rng.randrange(0, 37)
        a, b = g[p:p + 20], F._rc(g[p + 50:p + 70])
        m1.append(("p%d" % i, a, "I" * 20)); m2.append(("p%d" % i, b, "I" * 20))
    m1.append(("u0", flank(20), "I" * 20)); m2.append(("u0", flank(20), "I" * 20))          # and one that does not align
    f1, f2 = str(tmp_path / "m_1.fq"), str(tmp_path / "m_2.fq")
    F._write_fastq(f1, m1, 1)
    F._write_fastq(f2, m2, 2)
    for args in (["-v", "0", "-a", "-X", "200"], ["-v", "1", "-k", "20", "-X", "120"], ["-n", "1", "-l", "10", "-a", "-X", "150", "-S"]):
        ref, got = _both(args + (["--best"] if best else []), ["-x", base, "-1", f1, "-2", f2])
        assert ref.stdout.count(b"\n") > 60          # the case does what it says
        _check(ref, got, args)


def test_binary_drops_the_pair_when_one_mate_record_does_not_parse(tmp_path):
    """A mate record without a sequence (FASTA) or cut short takes its PAIR out of the run in the reference, which parses
    the two mates of a read id together (pat.cpp:96-127); the later pairs stay aligned with each other."""
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(400))
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(g.encode(), dtype=np.uint8)]], ["g"], base, ftab_chars=4, off_rate=2)
    pairs = []
    for i in range(9):
        p = rng.randrange(0, 300)
        pairs.append((g[p:p + 24], F._rc(g[p + 60:p + 84])))
    for bad1, bad2 in ((2, None), (None, 5), (1, 6), (3, 3)):
        f1, f2 = str(tmp_path / "m_1.fa"), str(tmp_path / "m_2.fa")
        with open(f1, "w") as a, open(f2, "w") as b:
            for i, (x, y) in enumerate(pairs):
                a.write(">p%d\n%s\n" % (i, "" if i == bad1 else x))
                b.write(">p%d\n%s\n" % (i, "" if i == bad2 else y))
        for args in (["-f", "-v", "1", "-X", "200"], ["-f", "-v", "0", "-X", "200", "--best", "-S"]):
            ref, got = _both(args, ["-x", base, "-1", f1, "-2", f2], ["--batch", "4"])
            _check(ref, got, (bad1, bad2, args))


@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_GPU_FUZZ_MIXED_SEEDS", "6"))))
def test_binary_on_a_tabbed_file_with_pairs_and_unpaired_reads(seed, tmp_path):
    """A --12 file may mix five-field (paired) and three-field (unpaired) records (TabbedPatternSource, pat.cpp:977-1127):
    the reference aligns the pairs with its paired aligner (V1 without --best, V2 with it), the unpaired reads with the
    stateful unpaired one, and writes everything in input order -- with tiny batches, -a / -k / -m, dumps and SAM."""
    rng = random.Random(700 + seed)
    g = "".join(rng.choice("ACGT") for _ in range(600))
    g2 = "".join(rng.choice("ACGT") for _ in range(300))
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(x.encode(), dtype=np.uint8)] for x in (g, g2)], ["g", "h"], base, ftab_chars=4, off_rate=2)
    lines = []
    for i in range(rng.randrange(6, 16)):
        src = rng.choice((g, g2))
        L = rng.choice([18, 24, 30])
        p = rng.randrange(0, len(src) - 160)
        a = list(src[p:p + L]); b = list(F._rc(src[p + 90:p + 90 + L]))
        for s_ in (a, b):
            for _ in range(rng.choice([0, 0, 1, 2])):
                s_[rng.randrange(L)] = rng.choice("ACGT")
        qa = "".join(rng.choice("!+5?IIII") for _ in range(L)); qb = "".join(rng.choice("!+5?IIII") for _ in range(L))
        kind = rng.random()
        if kind < 0.4:
            lines.append("u%d\t%s\t%s" % (i, "".join(a), qa))
        elif kind < 0.5:
            lines.append("s%d\tACG\tIII" % i)                   # shorter than 4: skipped with a warning
        else:
            lines.append("p%d\t%s\t%s\t%s\t%s" % (i, "".join(a), qa, "".join(b), qb))
    tab = str(tmp_path / "mixed.tab")
    with open(tab, "w") as f:
        f.write("\n".join(lines) + "\n")
    for _ in range(3):
        args = rng.choice([["-v", "1"], ["-n", "2", "-l", "10"], ["-v", "2", "--best"], ["-n", "1", "-l", "8", "--best", "-k", "3"], ["-v", "0", "-a"],
                           ["-v", "2", "-m", "1"], ["-v", "3"], ["-n", "2", "-l", "12", "-M", "1"]]) + rng.choice([["-X", "200"], ["-X", "150", "-I", "20"]]) + F.out_options(rng)
        if not F._args_ok(args):
            continue
        ref, got = _both(args, ["-x", base, "--12", tab], rng.choice([[], ["--batch", "3"], ["--batch", "5", "--inflight", "1"]]))
        _check(ref, got, (lines, args))


@pytest.mark.parametrize("fmt", ["--12", "--interleaved"])
def test_binary_reads_one_file_pairs_from_standard_input(fmt, tmp_path):
    """`--12 -` / `--interleaved -`: both mate streams need the input from its start, so the binary spools standard input
    to a temporary file; same output as the reference's on the same bytes."""
    rng = random.Random(11)
    g = "".join(rng.choice("ACGT") for _ in range(500))
    base = str(tmp_path / "g")
    EB.build_index([F.LUT[np.frombuffer(g.encode(), dtype=np.uint8)]], ["g"], base, ftab_chars=4, off_rate=2)
    recs = []
    for i in range(7):
        p = rng.randrange(0, 300)
        recs.append(("p%d" % i, g[p:p + 22], F._rc(g[p + 80:p + 102])))
    if fmt == "--12":
        data = "".join("%s\t%s\t%s\t%s\t%s\n" % (n, a, "I" * 22, b, "I" * 22) for n, a, b in recs)
    else:
        data = "".join("@%s/1\n%s\n+\n%s\n@%s/2\n%s\n+\n%s\n" % (n, a, "I" * 22, n, b, "I" * 22) for n, a, b in recs)
    for args in (["-v", "1", "-X", "200"], ["-v", "1", "-X", "200", "--best", "-S"]):
        ref = subprocess.run([F.REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, fmt, "-"], input=data.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        got = subprocess.run([BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, fmt, "-"], input=data.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert ref.stdout.count(b"\n") >= 10
        _check(ref, got, (fmt, args))
