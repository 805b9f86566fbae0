"""The rest of the `.ebwt` family: 64-bit `.ebwtl` builds, other-endian files and
bowtie2-build's `.bt2` side layout load to the same in-memory image as the plain index, and searches on
them give what the unmodified reference gives on the same files (tests/golden/family, made by
oracle/gen_golden_family.py: bowtie-build-l / bowtie-align-l outputs, and the record that bowtie-align-s
does not tell the re-written files from the original)."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
from functools import lru_cache

import numpy as np
import pytest

import common as T
import index_variants as V
import oracle_lib as OL
import refrun as R
from bowtie_amd import _abi as A
from bowtie_amd import aligner as AL

F = os.path.join(T.G, "family")
LARGE = os.path.join(F, "multi_l")


@lru_cache(maxsize=None)
def fam():
    with open(os.path.join(F, "MANIFEST.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def variants(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("variants"))
    V.write_swapped(os.path.join(T.G, "multi"), os.path.join(d, "multi_be"))
    V.write_bt2(os.path.join(T.G, "multi"), os.path.join(d, "multi_bt2"))
    return d


def _check(run, sam: bytes):
    with gzip.open(os.path.join(T.G, run["file"]), "rb") as f:
        want = f.read()
    got = T.strip_sam(sam)
    if got != want:
        for i, (a, b) in enumerate(zip(got.split(b"\n"), want.split(b"\n"))):
            if a != b:
                raise AssertionError("%s: first difference at SAM line %d:\n got  %r\n want %r" % (run["file"], i, a, b))
        raise AssertionError("%s: line counts differ" % run["file"])
    assert hashlib.md5(sam).hexdigest() == run["md5"], run["file"]


def _render(run, res, batch, refnames):
    pol = T.MODES[run["mode"]]
    return R.render(batch, res, refnames, sam=True, mhits=pol.get("mhits", 0xFFFFFFFF), sample_max=pol.get("sample_max", False))


def _render_pairs(run, res, b1, b2, refnames):
    pol = T.MODES[run["mode"]]
    return R.render_pairs(b1, b2, res, refnames, sam=True, mhits=pol.get("mhits", 0xFFFFFFFF))


# ---- loader: every member of the family is the same image -----------------------------------------------
def test_large_index_loads_to_the_small_index_image():
    for mirror in (False, True):
        small = AL.index_digest(os.path.join(T.G, "multi"), mirror)
        large = AL.index_digest(LARGE, mirror)
        assert small[0] == A.BT_INDEX_EBWT and large[0] == A.BT_INDEX_EBWTL
        assert small[1:] == large[1:]


def test_other_endian_and_bt2_files_load_to_the_same_image(variants):
    for mirror in (False, True):
        small = AL.index_digest(os.path.join(T.G, "multi"), mirror)
        be = AL.index_digest(os.path.join(variants, "multi_be"), mirror)
        b2 = AL.index_digest(os.path.join(variants, "multi_bt2"), mirror)
        assert be[0] == (A.BT_INDEX_EBWT | A.BT_INDEX_SWAPPED) and b2[0] == A.BT_INDEX_BT2
        assert be[1:] == small[1:] and b2[1:] == small[1:]
    assert (AL.restore_text(os.path.join(variants, "multi_bt2")) == T.joined_text("multi")).all()
    assert (AL.restore_text(LARGE) == T.joined_text("multi")).all()


def test_bt2_takes_precedence_like_adjustEbwtBase(variants, tmp_path):
    """<base>.1.bt2 next to <base>.1.ebwt: the reference opens the .bt2 (ebwt.cpp:43-46)."""
    d = str(tmp_path)
    for e in V.EXTS:
        shutil.copyfile(os.path.join(T.G, "e_coli.%s.ebwt" % e), os.path.join(d, "x.%s.ebwt" % e))
        shutil.copyfile(os.path.join(variants, "multi_bt2.%s.bt2" % e), os.path.join(d, "x.%s.bt2" % e))
    assert AL.index_digest(os.path.join(d, "x"))[1:] == AL.index_digest(os.path.join(T.G, "multi"))[1:]


def test_damaged_headers_are_refused(tmp_path):
    src = open(os.path.join(T.G, "multi.1.ebwt"), "rb").read()
    d = str(tmp_path)

    def load(b1: bytes):
        with open(os.path.join(d, "bad.1.ebwt"), "wb") as f:
            f.write(b1)
        shutil.copyfile(os.path.join(T.G, "multi.2.ebwt"), os.path.join(d, "bad.2.ebwt"))
        out = (AL.C.c_uint64 * 8)()
        return AL.lib().bt_index_digest(os.path.join(d, "bad").encode(), 0, out)

    assert load(src) == A.BT_OK
    assert load(src[:2000]) == A.BT_ERR_IO                                   # truncated
    assert load(b"\x02\0\0\0" + src[4:]) == A.BT_ERR_FORMAT                   # not an endianness mark
    b = bytearray(src); b[8:12] = (7).to_bytes(4, "little")                  # lineRate of the 64-bit build in a .ebwt
    assert load(bytes(b)) == A.BT_ERR_FORMAT
    b = bytearray(src); b[28:32] = (0xFFFFFFF0).to_bytes(4, "little")        # nPat beyond the text length
    assert load(bytes(b)) == A.BT_ERR_FORMAT
    b = bytearray(src); b[28:32] = (0).to_bytes(4, "little")
    assert load(bytes(b)) == A.BT_ERR_FORMAT
    n_pat = int.from_bytes(src[28:32], "little")
    o = 32 + 4 * n_pat
    b = bytearray(src); b[o:o + 4] = (0x7FFFFFFF).to_bytes(4, "little")      # nFrag
    assert load(bytes(b)) == A.BT_ERR_FORMAT
    b = bytearray(src); b[o + 4:o + 8] = (0x00FFFFFF).to_bytes(4, "little")  # first fragment starts past the end
    assert load(bytes(b)) == A.BT_ERR_FORMAT
    assert AL.lib().bt_index_digest(os.path.join(d, "nothing").encode(), 0, (AL.C.c_uint64 * 8)()) == A.BT_ERR_IO


# ---- the oracle restating bowtie-align-l -------------------------------------------------------------------
@lru_cache(maxsize=None)
def wide_oracle():
    return OL.OracleIndex(os.path.join(T.G, "multi"), wide=True)


@pytest.mark.parametrize("run", fam()["runs"], ids=lambda r: os.path.basename(r["file"])[:-7])
def test_oracle_wide_matches_bowtie_align_l(run):
    batch = T.read_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = R.oracle_search(wide_oracle(), OL.make_policy(**kw), batch, cap=T.hit_cap_for(kw))
    _check(run, _render(run, res, batch, wide_oracle().refnames))


@pytest.mark.parametrize("run", fam()["paired_runs"], ids=lambda r: os.path.basename(r["file"])[:-7])
def test_oracle_wide_paired_matches_bowtie_align_l(run):
    b1, b2 = T.pair_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = R.oracle_search_pairs(wide_oracle(), OL.make_policy(**kw), b1, b2, cap=2048 if kw.get("all_hits") else None)
    _check(run, _render_pairs(run, res, b1, b2, wide_oracle().refnames))


def test_the_64_bit_build_is_not_the_32_bit_build():
    """The fixtures do tell the two apart (a hit drawn from a range of rows with two 32-bit draws)."""
    assert sum(1 for r in fam()["runs"] if not r["same_as_small"]) >= 10


# ---- the per-read automaton, host build, on the converted image ------------------------------------------------
@pytest.mark.parametrize("run", [r for r in fam()["runs"] if r["reads"] in ("syn36", "syn150", "syn50lowq")],
                         ids=lambda r: os.path.basename(r["file"])[:-7])
def test_emu_on_large_index_matches_bowtie_align_l(run):
    import emu_lib as E
    emu = E.EmuAligner(LARGE)
    batch = T.read_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = emu.align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, n_lanes=37)
    _check(run, _render(run, res, batch, wide_oracle().refnames))


# ---- GPU --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def glarge():
    ix = AL.Index(LARGE)
    assert ix.info.variant == A.BT_INDEX_EBWTL
    return ix


@pytest.mark.gpu
@pytest.mark.parametrize("run", fam()["runs"], ids=lambda r: os.path.basename(r["file"])[:-7])
def test_gpu_large_index_matches_bowtie_align_l(run, glarge):
    batch = T.read_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = AL.Aligner(glarge, A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw))
    _check(run, _render(run, res, batch, glarge.refnames))


@pytest.mark.gpu
@pytest.mark.parametrize("run", fam()["paired_runs"], ids=lambda r: os.path.basename(r["file"])[:-7])
def test_gpu_large_index_paired_matches_bowtie_align_l(run, glarge):
    b1, b2 = T.pair_set("multi", run["reads"])
    kw = T.MODES[run["mode"]]
    res = AL.Aligner(glarge, A.make_policy(**kw)).align_pairs(b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    _check(run, _render_pairs(run, res, b1, b2, glarge.refnames))


@pytest.mark.gpu
@pytest.mark.parametrize("vname", ["multi_be", "multi_bt2"])
def test_gpu_other_endian_and_bt2_indexes_give_the_plain_index_results(vname, variants):
    """What the reference gives on these files is what it gives on the plain index (MANIFEST `variants`)."""
    ix = AL.Index(os.path.join(variants, vname))
    for v in [x for x in fam()["variants"] if x["variant"] == vname]:
        if v["mode"].startswith("pe_"):
            run = T.paired_runs("multi", [v["reads"]], [v["mode"]])[0]
            assert run["md5"] == v["md5"]
            b1, b2 = T.pair_set("multi", v["reads"])
            res = AL.Aligner(ix, A.make_policy(**T.MODES[v["mode"]])).align_pairs(b1, b2)
            T.check_pairs_against_golden(run, res, b1, b2, ix.refnames)
        else:
            run = T.golden_runs("multi", [v["reads"]], [v["mode"]])[0]
            assert run["md5"] == v["md5"]
            batch = T.read_set("multi", v["reads"])
            kw = T.MODES[v["mode"]]
            res = AL.Aligner(ix, A.make_policy(**kw)).align(batch, hit_cap=T.hit_cap_for(kw))
            T.check_against_golden(run, res, batch, ix.refnames)
    ix.close()


@pytest.mark.gpu
def test_cli_on_large_index_is_byte_identical_to_bowtie_align_l(tmp_path):
    """bowtie-amd -x multi_l: the SAM of the binary, end to end."""
    from bowtie_amd.synth import write_fastq
    binp = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")
    fq = str(tmp_path / "r.fq")
    for run in [r for r in fam()["runs"] if (r["reads"], r["mode"]) in (("syn36", "n2"), ("syn100", "v2"), ("syn50lowq", "n3_y"), ("syn36", "n2_best"))]:
        write_fastq(T.read_set("multi", run["reads"]), fq)
        p = subprocess.run([binp, "-S", "--sam-nohead"] + run["args"] + ["-x", LARGE, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert hashlib.md5(p.stdout).hexdigest() == run["md5"], run["file"]


def test_loader_survives_damaged_index_files(tmp_path):
    """tests/emu/index_asan.cpp: bt_host.cpp's loader (forward index, mirror, 2-bit reference) built with
    -fsanitize=address,undefined and fed 300 damaged copies of a small index -- flipped bits, truncations, huge or zero
    header words, zeroed files, trailing garbage.  It may load them or refuse them; it may not crash or trip a sanitizer."""
    import random
    import subprocess
    from bowtie_amd import ebwt_build as EB
    exe = str(tmp_path / "index_asan")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17", "-w",
                        "-o", exe, os.path.join(root, "tests", "emu", "index_asan.cpp"), os.path.join(root, "bowtie_amd", "csrc", "bt_host.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate("ACGT"):
        lut[ord(ch)] = i
    g = b"AGCATCGATCAGTATCTGACCNNNGTTAGGCATTACGGATCCATGCAAGTCTTGACGTACGGTCAATGC"
    EB.build_index([lut[np.frombuffer(g, dtype=np.uint8)], lut[np.frombuffer(b"ACGTTGCAAC", dtype=np.uint8)]], ["a x", "b"],
                   str(tmp_path / "g"), ftab_chars=3, off_rate=2)
    exts = ["1.ebwt", "2.ebwt", "3.ebwt", "4.ebwt", "rev.1.ebwt", "rev.2.ebwt"]
    orig = {e: open(str(tmp_path / "g") + "." + e, "rb").read() for e in exts}
    w = str(tmp_path / "w")
    seen = set()
    for seed in range(300):
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
        elif kind == "word" and len(b) >= 4:
            i = rng.randrange(0, min(len(b) - 3, 80))
            b[i:i + 4] = rng.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\xff\xff\xff\x7f", b"\x00\x00\x01\x00"])
        elif kind == "zero":
            b = bytearray(len(b))
        else:
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 50)))
        with open(w + "." + e, "wb") as f:
            f.write(b)
        p = subprocess.run([exe, w], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == 0 and b"ERROR" not in p.stderr and b"runtime error" not in p.stderr, (seed, e, kind, p.stderr.decode(errors="replace")[-800:])
        seen.add(p.stdout.split(b" len")[0])
    assert len(seen) >= 4          # loaded, and each of the three parts refused at least once


BUILD_L = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-build-l")


@pytest.mark.skipif(not os.path.exists(BUILD_L), reason="needs oracle/_ref/bowtie-build-l")
@pytest.mark.parametrize("seed", range(40))
def test_random_genomes_every_family_member_loads_to_the_same_image(seed, tmp_path):
    """Seeded random genomes (several sequences, N gaps, all-N sequences aside): the small index, bowtie-build-l's 64-bit
    index of the same FASTA, and the other-endian and .bt2-layout re-writes all load to one image (same digest of every
    array), forward and mirror, and the BWT inverts to the genome."""
    import random
    from bowtie_amd import ebwt_build as EB
    from test_ebwt_build import read_fa
    rng = random.Random(seed)
    recs = []
    for k in range(rng.randrange(1, 5)):
        L = rng.choice([8, 30, 100, 500, 2000])
        s = [rng.choice(rng.choice(["ACGT", "ACGT", "AC"])) for _ in range(L)]
        for _ in range(rng.choice([0, 0, 1, 2])):
            a = rng.randrange(0, L)
            b = min(L, a + rng.choice([1, 2, 5, 40]))
            s[a:b] = "N" * (b - a)
        if all(c == "N" for c in s):
            s[0] = "A"
        recs.append((">s%d words" % k, "".join(s)))
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as f:
        for h, s in recs:
            f.write(h + "\n" + s + "\n")
    off, ftab = rng.choice([1, 3, 5]), rng.choice([1, 3, 6])
    names, seqs = read_fa(fa)
    small = str(tmp_path / "small")
    EB.build_index(seqs, names, small, off_rate=off, ftab_chars=ftab)
    large = str(tmp_path / "large")
    subprocess.run([BUILD_L, "--offrate", str(off), "--ftabchars", str(ftab), "-q", fa, large], check=True, stderr=subprocess.DEVNULL)
    V.write_swapped(small, str(tmp_path / "be"))
    V.write_bt2(small, str(tmp_path / "bt2"))
    for mirror in (False, True):
        want = AL.index_digest(small, mirror)
        for other, variant in ((large, A.BT_INDEX_EBWTL), (str(tmp_path / "be"), A.BT_INDEX_EBWT | A.BT_INDEX_SWAPPED), (str(tmp_path / "bt2"), A.BT_INDEX_BT2)):
            got = AL.index_digest(other, mirror)
            assert got[0] == variant and got[1:] == want[1:], (other, mirror)
    text = np.concatenate([sq[sq != 4] for sq in seqs])
    for b in (small, large, str(tmp_path / "bt2")):
        assert (AL.restore_text(b) == text).all()
