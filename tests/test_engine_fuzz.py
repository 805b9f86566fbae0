"""Differential test of the search engines against the unmodified reference binary (oracle/_ref) on seeded random
small genomes -- several sequences, N gaps, repeats, lengths from 8 bases up, so that the '$' row, the eftab, fragment
ends and reads longer than the genome come up all the time -- with random reads and option sets: the reference's output
against (a) the oracle and (b) the host build of the device automatons (tests/emu), each between the C++ parser and the
C++ formatter.  Unpaired through both engines (default and --best / --strata / -M), pairs through --best.  No GPU."""
import os
import random
import subprocess

import numpy as np
import pytest

import cli_cases as CC
import common as T
import emu_lib as E
import oracle_lib as OL
import refrun as R
from bowtie_amd import _abi as A
from bowtie_amd import ebwt_build as EB
from bowtie_amd import hostio as H

REF_BIN = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-align-s")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="needs the reference binary (make -C oracle ref)")

def _seeds(n):
    """BT_FUZZ_OFFSET moves the window of seeds (one-off runs over fresh ones)."""
    o = int(os.environ.get("BT_FUZZ_OFFSET", "0"))
    return range(o, o + n)


LUT = np.full(256, 4, np.uint8)
for _i, _ch in enumerate("ACGT"):
    LUT[ord(_ch)] = _i


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def make_genome(rng):
    seqs = []
    for _ in range(rng.choice([1, 1, 2, 3])):
        L = rng.choice([8, 12, 20, 40, 80, 150, 300])
        alpha = rng.choice(["ACGT", "ACGT", "AC", "AAAC"])           # low-complexity genomes make many multi-mappers
        s = [rng.choice(alpha) for _ in range(L)]
        if L >= 40 and rng.random() < 0.5:                            # a repeat
            k = rng.randrange(8, L // 3)
            a, b = rng.randrange(0, L - k), rng.randrange(0, L - k)
            s[b:b + k] = s[a:a + k]
        if L >= 20 and rng.random() < 0.4:                            # an N gap: two fragments
            a = rng.randrange(2, L - 4)
            for i in range(a, min(L - 2, a + rng.randrange(1, 5))):
                s[i] = "N"
        seqs.append("".join(s))
    return seqs


def make_reads(rng, seqs, n, lens):
    out = []
    for i in range(n):
        L = rng.choice(lens)
        g = rng.choice(seqs)
        if len(g) >= L and rng.random() < 0.85:
            p = rng.randrange(0, len(g) - L + 1)
            s = list(g[p:p + L].replace("N", "A"))
            for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
                s[rng.randrange(L)] = rng.choice("ACGT")
            if rng.random() < 0.05:
                s[rng.randrange(L)] = "N"
            s = "".join(s)
            if rng.random() < 0.5:
                s = _rc(s.replace("N", "X")).replace("X", "N")
        else:
            s = "".join(rng.choice("ACGT") for _ in range(L))
        q = "".join(rng.choice("!+5?IIII") for _ in range(L))
        out.append((rng.choice(["r%d", "r%d", "r%d with words", "r%d/1", "read_%d\ttabbed"]) % i, s, q))
    return out


UNPAIRED_POLICIES = [
    ["-v", "0"], ["-v", "1"], ["-v", "2"], ["-n", "0", "-l", "6"], ["-n", "1", "-l", "8"], ["-n", "2", "-l", "10"],
    ["-n", "3", "-l", "12", "-e", "120"], ["-n", "2", "-l", "8", "-e", "40"], ["-n", "2", "-l", "8", "--nomaqround"],
    ["-v", "2", "--best"], ["-v", "3"], ["-n", "2", "-l", "8", "--best"], ["-n", "3", "-l", "10", "--best", "--strata", "-k", "3"],
    ["-v", "2", "--best", "--strata", "-m", "2", "-k", "2"], ["-n", "1", "-l", "6", "--best", "-M", "2"], ["-v", "1", "--best", "-M", "1"],
]
REPORTS = [[], [], ["-k", "3"], ["-a"], ["-a"], ["-m", "1"], ["-k", "2", "-m", "3"], ["--nofw"], ["--norc"], ["-a", "--maxbts", "5"]]


def out_options(rng):
    """A random set of output options: the formatters' columns and SAM fields against the reference's too."""
    if rng.random() < 0.5:
        o = ["-S"]
        r = rng.random()
        if r < 0.6:
            o += ["--sam-nohead"]
        else:                                                    # with the header: @HD, @SQ (or not), @RG, @PG
            if r < 0.75:
                o += ["--sam-nosq"]
            if r > 0.85:
                o += ["--sam-RG", "ID:g%d" % rng.randrange(9), "--sam-RG", "SM:x y"]
        for opt, p in ((["--mapq", str(rng.choice([0, 7, 40]))], 0.3), (["--no-unal"], 0.3), (["--fullref"], 0.2),
                       (["--sam-no-qname-trunc"], 0.2), (["--refidx"], 0.1)):
            if rng.random() < p:
                o += opt
        return o
    o = []
    for opt, p in ((["-B", str(rng.choice([1, 5]))], 0.3), (["--refidx"], 0.3), (["--fullref"], 0.2), (["--cost"], 0.3),
                   (["--showseed"], 0.2), (["--suppress", rng.choice(["1", "2,3", "5,6,7", "8"])], 0.3)):
        if rng.random() < p:
            o += opt
    return o



def _header_for(ref_stdout: bytes, oi, opts, ex) -> bytes:
    """The SAM header the formatter writes for this run; the @PG line quotes the command line, so the reference's own is
    handed in."""
    if not opts.sam or ex["sam_nohead"]:
        return b""
    head = [l for l in ref_stdout.split(b"\n") if l.startswith(b"@")]
    cl = head[-1].split(b'CL:"', 1)[1][:-1].decode()
    return H.sam_header(oi.refnames, oi.reflens, opts, cl, "\t".join(ex["rg"]) or None)


def _summary_of(stderr: bytes):
    return [l for l in stderr.decode(errors="replace").strip().split("\n") if l.startswith("#") or l.startswith("Reported") or l.startswith("No align")]


def _write_fastq(path, reads, mate=0):
    with open(path, "w") as f:
        for name, s, q in reads:
            f.write("@%s%s\n%s\n+\n%s\n" % (name, "/%d" % mate if mate else "", s, q))


def _args_ok(args):
    a = set(args)
    if "--strata" in a and not ({"-a", "-k", "-m", "-M"} & a):
        return False
    return True


def _policy(pol):
    return A.make_policy(**pol)


@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_SEEDS", "250"))))
def test_unpaired_engines_against_the_reference(seed, tmp_path):
    rng = random.Random(seed)
    seqs = make_genome(rng)
    base = str(tmp_path / "g")
    EB.build_index([LUT[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs], ["s%d some description" % i if i % 2 == 0 else "t%d" % i for i in range(len(seqs))], base,
                   ftab_chars=rng.choice([1, 2, 3, 4, 6]), off_rate=rng.choice([1, 2, 3, 5]))
    lens = [4, 5, 7, 10, 12, 16, 22, 30]
    if max(len(g) for g in seqs) >= 150 and rng.random() < 0.5:
        lens = [30, 60, 105, 110, 113, 130]        # either side of the kernel builds' read-length limits (104, 112)
    reads = make_reads(rng, seqs, rng.randrange(4, 14), lens)
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads)
    for _ in range(3):
        pol_args = rng.choice(UNPAIRED_POLICIES)
        rep = [x for x in rng.choice(REPORTS)]
        if "-M" in pol_args or "-m" in pol_args or ("-k" in pol_args and "-k" in rep):
            rep = [x for x in rep if x not in ("-m", "-k", "1", "2", "3")] if ("-M" in pol_args or "-m" in pol_args) else []
        args = pol_args + rep + out_options(rng) + ["--seed", str(rng.randrange(0, 5))]
        if not _args_ok(args):
            continue
        ref = subprocess.run([REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, fq],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if ref.returncode != 0:
            assert ref.returncode > 0 and (b"is less than" in ref.stderr or b"at least" in ref.stderr), (args, ref.stderr[-300:])
            continue                                   # a read shorter than the mode allows: the reference stops with an error
        rd, pol, out, ex = CC.interpret(args)
        b1 = H.read_all(fq, **rd)
        oi = OL.OracleIndex(base)
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        p = _policy(pol)
        per_o = R.oracle_search(oi, OL.make_policy(**pol), b1, cap=cap)
        per_e = E.EmuAligner(base).align(p, b1, hit_cap=cap, lite=(not p.best and rng.random() < 0.5), no_rl=(not p.best and rng.random() < 0.3))
        head = _header_for(ref.stdout, oi, opts, ex)
        for who, per in (("oracle", per_o), ("device automaton (host build)", per_e)):
            hits, nh, st, pool = H.pack_hits(per, cap)
            got, tally = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
            got = head + got
            assert got == ref.stdout, (who, seqs, args)
            assert H.summary(tally).strip().split("\n") == _summary_of(ref.stderr), (who, args)


PAIRED_POLICIES = [["-v", "0"], ["-v", "1"], ["-v", "2"], ["-n", "1", "-l", "8"], ["-n", "2", "-l", "10"], ["-v", "3"], ["-n", "3", "-l", "8", "-e", "100"]]
PAIRED_REPORTS = [[], ["-k", "2"], ["-a"], ["-m", "1"], ["-a", "--strata"], ["-M", "1"], ["--ff"], ["--rf"], ["--nofw"], ["--allow-contain"], ["--pairtries", "2"]]


@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_SEEDS", "120"))))
def test_paired_engine_against_the_reference(seed, tmp_path):
    _paired_fuzz(seed, tmp_path, best=True)


@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_SEEDS", "100"))))
def test_paired_engine_without_best_against_the_reference(seed, tmp_path):
    """PairedBWAlignerV1 (pairs without --best): oracle and the host build of bf_run_pair_v1 (the device code that has
    not run on a GPU yet)."""
    _paired_fuzz(seed, tmp_path, best=False)


def _paired_fuzz(seed, tmp_path, best):
    rng = random.Random((10_000 if best else 20_000) + seed)
    seqs = [s for s in make_genome(rng)]
    seqs.append("".join(rng.choice("ACGT") for _ in range(rng.choice([60, 120, 250]))))     # room for a fragment
    base = str(tmp_path / "g")
    EB.build_index([LUT[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs], ["s%d some description" % i if i % 2 == 0 else "t%d" % i for i in range(len(seqs))], base,
                   ftab_chars=rng.choice([1, 2, 4, 6]), off_rate=rng.choice([1, 3, 5]))
    m1, m2 = [], []
    for i in range(rng.randrange(3, 10)):
        g = rng.choice(seqs)
        L1, L2 = rng.choice([5, 8, 12, 20]), rng.choice([5, 8, 12, 20])
        F = rng.randrange(max(L1, L2), max(L1, L2) + 60)
        if len(g) >= F and rng.random() < 0.85:
            p = rng.randrange(0, len(g) - F + 1)
            frag = g[p:p + F].replace("N", "C")
            a, b = list(frag[:L1]), list(_rc(frag[F - L2:]))
            for s in (a, b):
                for _ in range(rng.choice([0, 0, 1, 2])):
                    s[rng.randrange(len(s))] = rng.choice("ACGT")
            a, b = "".join(a), "".join(b)
            if rng.random() < 0.3:
                a, b = b, a
        else:
            a = "".join(rng.choice("ACGT") for _ in range(L1)); b = "".join(rng.choice("ACGT") for _ in range(L2))
        m1.append(("p%d" % i, a, "".join(rng.choice("!+5?IIII") for _ in a)))
        m2.append(("p%d" % i, b, "".join(rng.choice("!+5?IIII") for _ in b)))
    f1, f2 = str(tmp_path / "m_1.fq"), str(tmp_path / "m_2.fq")
    _write_fastq(f1, m1, 1)
    _write_fastq(f2, m2, 2)
    for _ in range(3):
        args = rng.choice(PAIRED_POLICIES) + (["--best"] if best else []) + rng.choice(PAIRED_REPORTS) + \
            rng.choice([["-X", "100"], ["-X", "60", "-I", "10"], ["-X", "250"]]) + rng.choice([[], [], ["-5", "1"], ["-3", "2"], ["-5", "2", "-3", "1"]]) + out_options(rng)
        if not _args_ok(args):
            continue
        if not best and "--strata" in args and "-M" not in args and args[:2] != ["-v", "3"]:
            continue            # "--strata must be combined with --best" unless -v 3 / -M made the run stateful already
        ref = subprocess.run([REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, "-1", f1, "-2", f2],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if ref.returncode != 0:
            assert ref.returncode > 0 and (b"is less than" in ref.stderr or b"at least" in ref.stderr), (args, ref.stderr[-300:])
            continue
        rd, pol, out, ex = CC.interpret(args)
        b1, b2 = H.read_all(f1, mate=1, **rd), H.read_all(f2, mate=2, **rd)
        # the insert limits as the aligner sees them: less the trimming at the fragment's outer ends (aligner.h:1921-1935)
        o1 = rd.get("trim5", 0) if pol.get("mate1_fw", True) else rd.get("trim3", 0)
        o2 = rd.get("trim3", 0) if pol.get("mate2_fw", False) else rd.get("trim5", 0)
        pol = dict(pol, min_ins=max(0, max(0, pol.get("min_ins", 0) - o1) - o2), max_ins=max(0, max(0, pol.get("max_ins", 250) - o1) - o2))
        oi = OL.OracleIndex(base)
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else 2 * max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        if not best:
            pol = dict(pol, pe_v1=True)
        per_o = R.oracle_search_pairs(oi, OL.make_policy(**pol), b1, b2, cap=cap, v1=not best)
        per_e = E.EmuAligner(base).align_pairs(_policy(pol), b1, b2, hit_cap=cap)
        head = _header_for(ref.stdout, oi, opts, ex)
        for who, per in (("oracle", per_o), ("device automaton (host build)", per_e)):
            hits, nh, st, pool = H.pack_hits(per, cap)
            got, tally = H.format_pairs(b1, b2, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
            got = head + got
            assert got == ref.stdout, (who, seqs, args)
            assert H.summary(tally).strip().split("\n") == _summary_of(ref.stderr), (who, args)


MEDIUM_POLICIES = [
    ["-n", "2"], ["-n", "2", "-l", "28", "-e", "70"], ["-n", "3", "-l", "20", "-e", "140"], ["-n", "1", "-l", "36"], ["-n", "0"],
    ["-v", "2"], ["-v", "1"], ["-v", "3"], ["-n", "2", "-y"], ["-n", "2", "--maxbts", "20"], ["-n", "3", "--nomaqround", "-e", "90"],
    ["-n", "2", "--best"], ["-n", "3", "-l", "24", "--best", "--strata", "-k", "4"], ["-v", "2", "--best", "-M", "3"], ["-n", "2", "--best", "-m", "2"],
]


@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_MEDIUM_SEEDS", "4"))))
def test_unpaired_engines_on_medium_genomes_against_the_reference(seed, tmp_path):
    """The same on genomes of 2 to 50 kbp with repeat families and real read lengths (36 to 100 bases), at the reference's
    default index parameters: deeper backtracking, seed extension, the quality budget."""
    rng = random.Random(50_000 + seed)
    L = rng.choice([2000, 10000, 50000])
    fam = ["".join(rng.choice("ACGT") for _ in range(rng.choice([60, 150, 400]))) for _ in range(3)]
    s = [rng.choice("ACGT") for _ in range(L)]
    for _ in range(L // 600):
        f = list(rng.choice(fam))
        for _ in range(rng.choice([0, 1, 3, 8])):
            f[rng.randrange(len(f))] = rng.choice("ACGT")
        p = rng.randrange(0, L - len(f))
        s[p:p + len(f)] = f
    if rng.random() < 0.5:
        p = rng.randrange(100, L - 100)
        s[p:p + rng.choice([1, 10, 50])] = "N" * len(s[p:p + rng.choice([1, 10, 50])])
    seqs = ["".join(s)]
    if rng.random() < 0.5:
        seqs.append("".join(rng.choice("ACGT") for _ in range(rng.choice([300, 3000]))))
    base = str(tmp_path / "g")
    EB.build_index([LUT[np.frombuffer(x.encode(), dtype=np.uint8)] for x in seqs], ["chr%d desc" % i for i in range(len(seqs))], base,
                   ftab_chars=rng.choice([6, 8, 10]), off_rate=rng.choice([3, 5]))
    rl = rng.choice([36, 50, 76, 100])
    reads = []
    for i in range(rng.randrange(15, 40)):
        g = rng.choice(seqs)
        p = rng.randrange(0, len(g) - rl)
        r = list(g[p:p + rl].replace("N", "A"))
        for _ in range(rng.choice([0, 1, 2, 2, 3, 4, 6])):
            r[rng.randrange(rl)] = rng.choice("ACGT")
        r = "".join(r)
        if rng.random() < 0.5:
            r = _rc(r)
        q = "".join(rng.choice("#+5:?DIIII") for _ in range(rl))
        reads.append(("r%d" % i, r, q))
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads)
    for _ in range(2):
        pol_args = rng.choice(MEDIUM_POLICIES)
        rep = rng.choice([[], [], ["-k", "3"], ["-a"], ["-m", "1"], ["--nofw"], ["--norc"]])
        if "-M" in pol_args or "-m" in pol_args or "-k" in pol_args:
            rep = [x for x in rep if x in ("--nofw", "--norc")]
        args = pol_args + rep + out_options(rng) + ["--seed", str(rng.randrange(0, 3))]
        if not _args_ok(args):
            continue
        ref = subprocess.run([REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, fq],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert ref.returncode == 0, (args, ref.stderr[-300:])
        rd, pol, out, ex = CC.interpret(args)
        b1 = H.read_all(fq, **rd)
        oi = OL.OracleIndex(base)
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        p = _policy(pol)
        per_o = R.oracle_search(oi, OL.make_policy(**pol), b1, cap=cap)
        per_e = E.EmuAligner(base).align(p, b1, hit_cap=cap, lite=(not p.best and rl <= 100 and rng.random() < 0.5),
                                         no_rl=(not p.best and rng.random() < 0.3), pal_cap=16384)
        head = _header_for(ref.stdout, oi, opts, ex)
        for who, per in (("oracle", per_o), ("device automaton (host build)", per_e)):
            hits, nh, st, pool = H.pack_hits(per, cap)
            got, _ = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
            assert head + got == ref.stdout, (who, args)


def _medium_genome(rng):
    L = rng.choice([2000, 10000, 50000])
    fam = ["".join(rng.choice("ACGT") for _ in range(rng.choice([60, 150, 400]))) for _ in range(3)]
    s = [rng.choice("ACGT") for _ in range(L)]
    for _ in range(L // 600):
        f = list(rng.choice(fam))
        for _ in range(rng.choice([0, 1, 3, 8])):
            f[rng.randrange(len(f))] = rng.choice("ACGT")
        p = rng.randrange(0, L - len(f))
        s[p:p + len(f)] = f
    seqs = ["".join(s)]
    if rng.random() < 0.5:
        seqs.append("".join(rng.choice("ACGT") for _ in range(rng.choice([600, 3000]))))
    return seqs


@pytest.mark.parametrize("best", [True, False], ids=["best", "without_best"])
@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_MEDIUM_SEEDS", "3"))))
def test_paired_engines_on_medium_genomes_against_the_reference(seed, best, tmp_path):
    rng = random.Random(70_000 + seed)
    seqs = _medium_genome(rng)
    base = str(tmp_path / "g")
    EB.build_index([LUT[np.frombuffer(x.encode(), dtype=np.uint8)] for x in seqs], ["chr%d desc" % i for i in range(len(seqs))], base,
                   ftab_chars=rng.choice([6, 8, 10]), off_rate=rng.choice([3, 5]))
    rl = rng.choice([36, 50, 75])
    m1, m2 = [], []
    for i in range(rng.randrange(10, 30)):
        g = rng.choice(seqs)
        F = rng.randrange(max(rl, 120), min(len(g), 450))
        p = rng.randrange(0, len(g) - F + 1)
        frag = g[p:p + F].replace("N", "C")
        a, b = list(frag[:rl]), list(_rc(frag[F - rl:]))
        for x in (a, b):
            for _ in range(rng.choice([0, 0, 1, 2, 3])):
                x[rng.randrange(rl)] = rng.choice("ACGT")
        a, b = "".join(a), "".join(b)
        if rng.random() < 0.2:
            a, b = b, a
        m1.append(("p%d" % i, a, "".join(rng.choice("#+5:?DIIII") for _ in a)))
        m2.append(("p%d" % i, b, "".join(rng.choice("#+5:?DIIII") for _ in b)))
    f1, f2 = str(tmp_path / "m_1.fq"), str(tmp_path / "m_2.fq")
    _write_fastq(f1, m1, 1)
    _write_fastq(f2, m2, 2)
    pols = [["-n", "2"], ["-n", "1", "-l", "30"], ["-v", "2"], ["-v", "0"], ["-n", "3", "-e", "120"], ["-v", "1"]] + ([["-v", "3"]] if best else [])
    reps = [[], ["-k", "2"], ["-a"], ["-m", "1"], ["--ff"], ["--nofw"], ["--pairtries", "3"]] + ([["-M", "1"], ["-a", "--strata"]] if best else [])
    for _ in range(2):
        args = rng.choice(pols) + (["--best"] if best else []) + rng.choice(reps) + rng.choice([["-X", "500"], ["-X", "300", "-I", "100"], []]) + out_options(rng)
        if not _args_ok(args):
            continue
        if not best and "--strata" in args and "-M" not in args and args[:2] != ["-v", "3"]:
            continue            # "--strata must be combined with --best" unless -v 3 / -M made the run stateful already
        ref = subprocess.run([REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", base, "-1", f1, "-2", f2],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert ref.returncode == 0, (args, ref.stderr[-300:])
        rd, pol, out, ex = CC.interpret(args)
        if not best:
            pol = dict(pol, pe_v1=True)
        b1, b2 = H.read_all(f1, mate=1, **rd), H.read_all(f2, mate=2, **rd)
        oi = OL.OracleIndex(base)
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else 2 * max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        per_o = R.oracle_search_pairs(oi, OL.make_policy(**pol), b1, b2, cap=cap, v1=not best)
        per_e = E.EmuAligner(base).align_pairs(_policy(pol), b1, b2, hit_cap=cap)
        head = _header_for(ref.stdout, oi, opts, ex)
        for who, per in (("oracle", per_o), ("device automaton (host build)", per_e)):
            hits, nh, st, pool = H.pack_hits(per, cap)
            got, _ = H.format_pairs(b1, b2, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
            assert head + got == ref.stdout, (who, args)


REF_L = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-align-l")
BUILD_L = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-build-l")


@pytest.mark.skipif(not (os.path.exists(REF_L) and os.path.exists(BUILD_L)), reason="needs the 64-bit reference binaries (make -C oracle ref)")
@pytest.mark.parametrize("seed", _seeds(int(os.environ.get("BT_FUZZ_SEEDS", "60"))))
def test_the_64_bit_build_against_bowtie_align_l(seed, tmp_path):
    """bowtie-align-l on bowtie-build-l's index of the same random genomes (its two visible differences from the 32-bit
    build: two generator draws per reported row range, a smaller branch pool): against the oracle in its 64-bit mode on
    the small index, and against the host build of the device automatons on the .ebwtl files as the loader converts them."""
    rng = random.Random(30_000 + seed)
    seqs = make_genome(rng)
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        for i, sq in enumerate(seqs):
            f.write(">s%d words\n%s\n" % (i, sq))
    off, ftab = rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 3, 4, 6])
    small, large = str(tmp_path / "small"), str(tmp_path / "large")
    EB.build_index([LUT[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs], ["s%d words" % i for i in range(len(seqs))], small, ftab_chars=ftab, off_rate=off)
    subprocess.run([BUILD_L, "--offrate", str(off), "--ftabchars", str(ftab), "-q", fa, large], check=True, stderr=subprocess.DEVNULL)
    reads = make_reads(rng, seqs, rng.randrange(4, 14), [4, 5, 7, 10, 12, 16, 22, 30])
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads)
    oi = OL.OracleIndex(small, wide=True)
    for _ in range(3):
        pol_args = rng.choice(UNPAIRED_POLICIES)
        rep = [x for x in rng.choice(REPORTS)]
        if "-M" in pol_args or "-m" in pol_args or ("-k" in pol_args and "-k" in rep):
            rep = [x for x in rep if x not in ("-m", "-k", "1", "2", "3")] if ("-M" in pol_args or "-m" in pol_args) else []
        args = pol_args + rep + out_options(rng) + ["--seed", str(rng.randrange(0, 5))]
        if not _args_ok(args):
            continue
        ref = subprocess.run([REF_L, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", large, fq],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if ref.returncode != 0:
            assert ref.returncode > 0 and (b"is less than" in ref.stderr or b"at least" in ref.stderr), (args, ref.stderr[-300:])
            continue
        rd, pol, out, ex = CC.interpret(args)
        b1 = H.read_all(fq, **rd)
        opts = H.out_opts(**out)
        cap = 4096 if pol.get("all_hits") else max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
        p = _policy(pol)
        per_o = R.oracle_search(oi, OL.make_policy(**pol), b1, cap=cap)
        per_e = E.EmuAligner(large).align(p, b1, hit_cap=cap, lite=(not p.best and rng.random() < 0.5))
        head = _header_for(ref.stdout, oi, opts, ex)
        for who, per in (("oracle (64-bit mode)", per_o), ("device automaton on the .ebwtl index", per_e)):
            hits, nh, st, pool = H.pack_hits(per, cap)
            got, tally = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
            got = head + got
            assert got == ref.stdout, (who, seqs, args)
            assert H.summary(tally).strip().split("\n") == _summary_of(ref.stderr), (who, args)
