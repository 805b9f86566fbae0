"""Shared test plumbing: golden fixtures, read sets, policy matrix, result comparison."""
from __future__ import annotations

import gzip
import hashlib
import json
import os
from functools import lru_cache

import numpy as np

import oracle_lib as OL
import refrun as R
from bowtie_amd import _abi as A
from bowtie_amd.reads import pack_reads, parse_fastq
from bowtie_amd.synth import synth_reads
from best_modes import PAIRED_V1_MODES  # noqa: E402
from best_modes import BEST_MODES, PAIRED_MODES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

# mode name -> policy kwargs (same matrix as oracle/gen_golden.py MODES)
MODES = {
    "v0": dict(mode="v", mms=0), "v1": dict(mode="v", mms=1), "v2": dict(mode="v", mms=2),
    "n0": dict(mode="n", mms=0), "n1": dict(mode="n", mms=1), "n2": dict(mode="n", mms=2),
    "n3": dict(mode="n", mms=3),
    "n2_k3": dict(mode="n", mms=2, khits=3), "v2_a": dict(mode="v", mms=2, all_hits=True),
    "n2_m1": dict(mode="n", mms=2, mhits=1),
    "n2_l20_e100": dict(mode="n", mms=2, seed_len=20, qual_thresh=100),
    "n2_nofw": dict(mode="n", mms=2, nofw=True), "v2_norc": dict(mode="v", mms=2, norc=True),
    "n2_nomaq": dict(mode="n", mms=2, maq_round=False),
    "n3_l22_e140_k2": dict(mode="n", mms=3, seed_len=22, qual_thresh=140, khits=2),
    "v1_k5": dict(mode="v", mms=1, khits=5), "n1_a_m20": dict(mode="n", mms=1, all_hits=True, mhits=20),
    "n3_a": dict(mode="n", mms=3, all_hits=True), "n2_e200_nomaq": dict(mode="n", mms=2, qual_thresh=200, maq_round=False),
    "n2_l12": dict(mode="n", mms=2, seed_len=12), "n2_maxbts10": dict(mode="n", mms=2, max_bts=10),
    "n3_y": dict(mode="n", mms=3, max_bts=0x7FFFFFFF), "v2_k100": dict(mode="v", mms=2, khits=100),
    "n0_a_m5": dict(mode="n", mms=0, all_hits=True, mhits=5), "n1_l36_e40": dict(mode="n", mms=1, seed_len=36, qual_thresh=40),
    "v2_nofw_k3": dict(mode="v", mms=2, nofw=True, khits=3),
}
MODES.update({k: v[1] for k, v in BEST_MODES.items()})
MODES.update({k: v[1] for k, v in PAIRED_MODES.items()})
MODES.update({k: v[1] for k, v in PAIRED_V1_MODES.items()})      # paired-end without --best (PairedBWAlignerV1)


@lru_cache(maxsize=None)
def manifest():
    with open(os.path.join(G, "MANIFEST.json")) as f:
        return json.load(f)


@lru_cache(maxsize=None)
def pair_set(index: str, name: str):
    """The read pairs oracle/gen_golden.py fed to the reference: (mates 1, mates 2)."""
    if name == "e_coli_1000_pe":
        return (pack_reads(parse_fastq(os.path.join(G, "e_coli_1000_1.fq"))),
                pack_reads(parse_fastq(os.path.join(G, "e_coli_1000_2.fq"))))
    from bowtie_amd.synth import synth_pairs
    L = int(name[2:])
    return synth_pairs(joined_text(index), 400, L, seed=L)


def paired_runs(index=None, reads=None, modes=None):
    out = []
    for r in manifest().get("paired_runs", []):
        if index and r["index"] != index:
            continue
        if reads and r["reads"] not in reads:
            continue
        if modes and r["mode"] not in modes:
            continue
        out.append(r)
    return out


def check_pairs_against_golden(run: dict, per_pair, b1, b2, refnames):
    pol = MODES[run["mode"]]
    sam = R.render_pairs(b1, b2, per_pair, refnames, sam=True, mhits=pol.get("mhits", 0xFFFFFFFF),
                         sample_max=bool(pol.get("sample_max")))
    with gzip.open(os.path.join(G, run["file"]), "rb") as f:
        want = f.read()
    got = strip_sam(sam)
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            if a != b:
                raise AssertionError("%s: first difference at SAM line %d:\n got  %r\n want %r" % (run["file"], i, a, b))
        raise AssertionError("%s: %d lines vs %d" % (run["file"], len(gl), len(wl)))
    assert hashlib.md5(sam).hexdigest() == run["md5"], run["file"] + ": full-SAM md5 differs"


def oracle_pair_results(index: str, b1, b2, kw, cap=None, counts=None, v1=False):
    pol = OL.make_policy(**kw)
    return R.oracle_search_pairs(oracle_index(index), pol, b1, b2, cap=cap, counts=counts, v1=v1)


@lru_cache(maxsize=None)
def pe_v1_manifest():
    with open(os.path.join(G, "pe_v1", "MANIFEST.json")) as f:
        return json.load(f)


def paired_v1_runs(index=None, reads=None, modes=None):
    """tests/golden/pe_v1 (oracle/gen_golden_pe_v1.py): the reference's paired-end output without --best."""
    return [r for r in pe_v1_manifest()["runs"]
            if (not index or r["index"] == index) and (not reads or r["reads"] in reads) and (not modes or r["mode"] in modes)]


def golden_runs(index=None, reads=None, modes=None):
    out = []
    for r in manifest()["runs"]:
        if "error" in r:
            continue
        if index and r["index"] != index:
            continue
        if reads and r["reads"] not in reads:
            continue
        if modes and r["mode"] not in modes:
            continue
        out.append(r)
    return out


@lru_cache(maxsize=None)
def oracle_index(name: str) -> OL.OracleIndex:
    return OL.OracleIndex(os.path.join(G, name))


@lru_cache(maxsize=None)
def joined_text(name: str) -> np.ndarray:
    return oracle_index(name).joined_text()


@lru_cache(maxsize=None)
def read_set(index: str, rname: str):
    """The same seeded read sets oracle/gen_golden.py fed to the reference."""
    if rname == "e_coli_1000":
        return pack_reads(parse_fastq(os.path.join(G, "e_coli_1000.fq")))
    text = joined_text(index)
    n = 400 if index == "e_coli" else 600
    if rname == "syn36":
        return synth_reads(text, n, 36, mm_dist=(0, 0, 1, 1, 2, 3), seed=36)
    if rname == "syn76":
        return synth_reads(text, n, 76, mm_dist=(0, 0, 1, 1, 2, 3), seed=76)
    if rname == "syn100":
        return synth_reads(text, n, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=100)
    if rname == "syn50lowq":
        return synth_reads(text, n, 50, mm_dist=(0, 1, 2, 3, 5), seed=50, lowq_frac=0.15, n_frac=0.05)
    if rname == "syn12":
        return synth_reads(text, n // 2, 12, mm_dist=(0, 0, 1), seed=12, n_frac=0.03)
    if rname == "syn150":
        return synth_reads(text, n // 2, 150, mm_dist=(0, 1, 2, 3, 5), seed=150, lowq_frac=0.05, n_frac=0.02)
    if rname == "syn110":
        return synth_reads(text, n // 2, 110, mm_dist=(0, 1, 2, 3), seed=110)
    raise KeyError(rname)


def strip_sam(sam: bytes) -> bytes:
    out = []
    for line in sam.split(b"\n"):
        if not line:
            continue
        c = line.split(b"\t")
        c[9] = b"*"
        c[10] = b"*"
        out.append(b"\t".join(c))
    return b"\n".join(out) + (b"\n" if out else b"")


def check_against_golden(run: dict, per_read, batch, refnames):
    """per_read from any backend -> SAM text; must equal the reference's output byte for byte
    (md5 of the full text) and field for field (stripped fixture)."""
    pol = MODES[run["mode"]]
    sam = R.render(batch, per_read, refnames, sam=True, mhits=pol.get("mhits", 0xFFFFFFFF),
                   sample_max=pol.get("sample_max", False))
    with gzip.open(os.path.join(G, run["file"]), "rb") as f:
        want = f.read()
    got = strip_sam(sam)
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            if a != b:
                raise AssertionError("%s: first difference at SAM line %d:\n got  %r\n want %r" %
                                     (run["file"], i, a, b))
        raise AssertionError("%s: %d lines vs %d" % (run["file"], len(gl), len(wl)))
    assert hashlib.md5(sam).hexdigest() == run["md5"], run["file"] + ": full-SAM md5 differs"


def compare_results(got, want, what=""):
    """Exact equality of (hits, hitsForThisRead, status) per read, with a readable first diff."""
    assert len(got) == len(want)
    bad = [i for i in range(len(got)) if got[i] != want[i]]
    if bad:
        i = bad[0]
        raise AssertionError("%s: %d/%d reads differ; first: read %d\n got  %r\n want %r" %
                             (what, len(bad), len(got), i, got[i], want[i]))


def oracle_results(index: str, batch, kw, cap=None, counts=None):
    pol = OL.make_policy(**kw)
    return R.oracle_search(oracle_index(index), pol, batch, cap=cap, counts=counts)


def result_digest(per_read) -> str:
    """Order-sensitive checksum of a result list (size-independent property checks)."""
    h = hashlib.sha256()
    for hits, tot, st in per_read:
        h.update(np.uint32(tot).tobytes() + bytes([st & 0xff]))
        for x in hits:
            h.update(np.array([x.tidx, x.toff, x.oms, x.cost, x.stratum, int(x.fw)], dtype=np.uint32).tobytes())
            for p, c in x.mms:
                h.update(bytes([p & 0xff, p >> 8, c]))
    return h.hexdigest()


def hit_cap_for(kw) -> int:
    """Hit slots per read large enough that no golden case overflows (-a on the tandem repeat)."""
    return 1024 if kw.get("all_hits") else max(64, int(kw.get("khits", 1)))   # >= any -M value used


def check_op_counts(oc, gc, what=""):
    """The product's op counters against the oracle's (= the reference algorithm's): equal -- with one allowance.  A
    mapLFEx on a one-row range that locus mode decided by the text is tallied as "both rows in one side pair" (they are
    neighbours; the row itself is not known there), which is wrong for one such row in 448: same_pair may exceed the
    oracle's by a small part of loc_lfex, never fall below it."""
    for f in ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames"):
        assert getattr(oc, f) == getattr(gc, f), (what, f, getattr(oc, f), getattr(gc, f))
    extra = int(gc.same_pair) - int(oc.same_pair)
    loc = int(getattr(gc, "loc_lfex", 0))
    assert 0 <= extra <= max(2, loc // 40), (what, "same_pair", int(oc.same_pair), int(gc.same_pair), loc)
