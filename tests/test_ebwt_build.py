"""bowtie_amd/ebwt_build.py (the GPU index synthesiser that feeds the hg19-scale benchmark) must
write exactly what reference bowtie-build writes.  The committed fixtures tests/golden/e_coli.* and
multi.* ARE reference bowtie-build outputs, so byte equality with them pins the builder without
needing the reference binary; when the binary is present a fresh random genome is checked too."""
import os
import subprocess

import numpy as np
import pytest
import torch

import common as T
import refrun as R
from bowtie_amd import ebwt_build as EB

EXTS = ("1.ebwt", "2.ebwt", "3.ebwt", "4.ebwt", "rev.1.ebwt", "rev.2.ebwt")


def read_fa(path):
    names, seqs, cur = [], [], []
    for line in open(path):
        line = line.rstrip()
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:])
            cur = []
        else:
            cur.append(line)
    seqs.append("".join(cur))
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate("ACGT"):
        lut[ord(c)] = i
        lut[ord(c.lower())] = i
    return names, [lut[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs]


def same(a, b):
    return all(open(a + "." + e, "rb").read() == open(b + "." + e, "rb").read() for e in EXTS)


def test_rebuild_multi_fixture(tmp_path):
    names, seqs = read_fa(os.path.join(T.G, "multi.fa"))
    EB.build_index(seqs, names, str(tmp_path / "m"), off_rate=3, ftab_chars=6)
    assert same(str(tmp_path / "m"), os.path.join(T.G, "multi"))


def test_rebuild_e_coli_fixture_chunked(tmp_path, monkeypatch):
    """Also drives the chunked giant-array passes and the sub-bucket sort split."""
    monkeypatch.setattr(EB, "_CH", 1_000_003)
    monkeypatch.setattr(EB, "_MAX_SORT", 200_000)
    text = T.joined_text("e_coli")
    EB.build_index([text], [T.oracle_index("e_coli").refnames[0]], str(tmp_path / "e"), off_rate=5, ftab_chars=7)
    assert same(str(tmp_path / "e"), os.path.join(T.G, "e_coli"))


@pytest.mark.skipif(not R.have_ref_binary(), reason="needs oracle/_ref/bowtie-build-s")
def test_random_genome_vs_reference_build(tmp_path):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, size=150000)
    rep = rng.integers(0, 4, size=700)
    for p in (1000, 50000, 120000, 149000):
        base[p:p + 700] = rep
    base[60000:60300] = 3
    base[-40:] = 3
    base[30000:30010] = 4
    fa = tmp_path / "r.fa"
    with open(fa, "w") as f:
        f.write(">a desc\n" + "".join("ACGTN"[c] for c in base) + "\n")
        f.write(">b\n" + "ACGTTTTTTTTTTTTTTTT" * 3 + "\n")
        f.write(">c\nNNNNACGTACGTACGTNNNN" + "T" * 43 + "NN\n")
        f.write(">d\n" + "T" * 64 + "\n")
    for offrate, ftab in ((5, 10), (2, 4)):
        subprocess.run([os.path.join(T.ROOT, "oracle", "_ref", "bowtie-build-s"), "--offrate", str(offrate),
                        "--ftabchars", str(ftab), "-q", str(fa), str(tmp_path / "ref")], check=True)
        names, seqs = read_fa(str(fa))
        EB.build_index(seqs, names, str(tmp_path / "mine"), off_rate=offrate, ftab_chars=ftab)
        assert same(str(tmp_path / "mine"), str(tmp_path / "ref"))


@pytest.mark.skipif(not R.have_ref_binary(), reason="needs oracle/_ref/bowtie-build-s")
@pytest.mark.parametrize("seed", range(int(os.environ.get("BT_FUZZ_SEEDS", "60"))))
def test_random_genomes_vs_reference_build(seed, tmp_path):
    """Seeded random genomes -- 1 to 6 sequences of 1 to 6000 bases, N runs anywhere (leading, trailing, whole
    sequences), repeats, homopolymers, lower case -- at random --offrate / --ftabchars: all six files byte for byte."""
    import random
    rng = random.Random(seed)
    recs = []
    for k in range(rng.randrange(1, 7)):
        L = rng.choice([1, 3, 8, 30, 100, 500, 2000, 6000])
        alpha = rng.choice(["ACGT", "ACGT", "ACGT", "AC", "T", "ACGTN"])
        s = [rng.choice(alpha) for _ in range(L)]
        for _ in range(rng.choice([0, 0, 1, 3])):
            a = rng.randrange(0, L); b = min(L, a + rng.choice([1, 2, 5, 40]))
            s[a:b] = "N" * (b - a)
        if L >= 100 and rng.random() < 0.5:
            k2 = rng.randrange(10, L // 3); a = rng.randrange(0, L - k2); b = rng.randrange(0, L - k2)
            s[b:b + k2] = s[a:a + k2]
        if rng.random() < 0.1:
            s = ["N"] * L                                            # a sequence with nothing in it
        s = "".join(s)
        if rng.random() < 0.2:
            s = s.lower()
        recs.append((">seq%d some description" % k if rng.random() < 0.5 else ">s%d" % k, s))
    if all(set(s.upper()) <= {"N"} for _, s in recs):
        recs.append((">last", "ACGTACGTAC"))
    fa = tmp_path / "r.fa"
    with open(fa, "w") as f:
        for h, s in recs:
            f.write(h + "\n")
            w = rng.choice([60, 70, 10**9])
            for i in range(0, len(s), w):
                f.write(s[i:i + w] + "\n")
    offrate, ftab = rng.choice([1, 2, 3, 5, 7]), rng.choice([1, 2, 4, 6, 8])
    subprocess.run([os.path.join(T.ROOT, "oracle", "_ref", "bowtie-build-s"), "--offrate", str(offrate),
                    "--ftabchars", str(ftab), "-q", str(fa), str(tmp_path / "ref")], check=True)
    names, seqs = read_fa(str(fa))
    EB.build_index(seqs, names, str(tmp_path / "mine"), off_rate=offrate, ftab_chars=ftab)
    for e in EXTS:
        assert open(str(tmp_path / "mine") + "." + e, "rb").read() == open(str(tmp_path / "ref") + "." + e, "rb").read(), e


def test_synthetic_genome_index_is_searchable(tmp_path):
    """ensure_big_index at toy scale: the oracle finds planted reads where they came from."""
    import oracle_lib as OL
    from bowtie_amd.synth import synth_reads
    base, text, note = EB.ensure_big_index(400_000, torch.device("cpu"), cache_dir=str(tmp_path))
    oi = OL.OracleIndex(base)
    assert (oi.joined_text() == text).all()
    assert oi.fw.nFrag == 48 and oi.fw.nPat == 24
    batch = synth_reads(text, 200, 50, mm_dist=(0,), seed=3, n_frac=0.0)
    res = R.oracle_search(oi, OL.make_policy("v", 0), batch)
    assert sum(1 for h, _, _ in res if h) >= 190         # reads straddling a fragment boundary are rejected


def test_locus_image_chain_by_chain_equals_the_one_walk_over_the_text(tmp_path):
    """The GPU loader derives the locus image's suffix array, reversed text and walk lengths a CHAIN at a time since round 6
    (bt_rank.h: bt_loc_chain -- from every sampled row down to the next one; round 5 walked from every row to a sampled one:
    2^offRate / 2 ranks per row instead of one).  The same function on the host against bt_loc_build_host's single walk over
    the text, on genomes of 1 .. 300 bases and every sampling rate 0 .. 6: the '$' row sampled or not, the last row sampled
    or not, chains of length one."""
    import ctypes as C
    import emu_lib as E
    from bowtie_amd import ebwt_build as EB
    rng = np.random.default_rng(12)
    n_checked = 0
    for L in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 200, 255, 256, 300]:
        for off_rate in (0, 1, 2, 3, 5, 6):
            seq = rng.integers(0, 4, size=L).astype(np.uint8)
            if L > 20 and off_rate == 2:
                seq[L // 2:L // 2 + 3] = 4                                   # two fragments
            base = str(tmp_path / ("g%d_%d" % (L, off_rate)))
            EB.build_index([seq], ["s"], base, off_rate=off_rate, ftab_chars=min(3, max(1, L // 4)) if L < 12 else 4)
            e = E.EmuAligner(base)
            e.L.emu_locus_chains_check.argtypes = [C.c_void_p, C.c_int]
            for mirror in (0, 1):
                assert e.L.emu_locus_chains_check(e.h, mirror) == 0, (L, off_rate, mirror)
                n_checked += 1
    assert n_checked > 500
