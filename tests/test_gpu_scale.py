"""Parity against the unmodified reference binary on an index too large for a fixture: a 120 Mbp synthetic
genome (48 fragments, interspersed repeats) indexed on the GPU (bowtie_amd/ebwt_build.py, files
byte-compatible with bowtie-build), the same files searched by `oracle/_ref/bowtie-align-s` on the box's
host cores and by the HIP path; the two SAM outputs are compared line by line.  Needs the reference binary
(`make -C oracle ref` in the build container; it travels to the GPU box with the snapshot) -- skipped
otherwise.  bench.py does the same at the 2.86 Gbp size on every default run (`diff_mismatches`)."""
import os
import subprocess

import pytest

import common as T
from bowtie_amd import _abi as A
from bowtie_amd import aligner as AL

REF_BIN = os.path.join(T.ROOT, "oracle", "_ref", "bowtie-align-s")
GENOME_BP = 120_000_000

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/bowtie-align-s not built")]


@pytest.fixture(scope="module")
def scale(tmp_path_factory):
    import torch
    from bowtie_amd import ebwt_build as EB
    d = str(tmp_path_factory.mktemp("idx120m"))
    base, text_np, note = EB.ensure_big_index(GENOME_BP, torch.device("cuda", 0), cache_dir=d)
    idx = AL.Index(base)
    assert idx.info.len >= 100_000_000 and idx.info.n_frag > 24
    yield dict(base=base, text=text_np, idx=idx, dir=d)
    idx.close()


def _reference_sam(base, args, inputs, out):
    # one thread: its output order is the input order (and the reference's --reorder with several threads does not
    # return on runs that suppress reads with -m: observed on this very case, bowtie-align-s -p 4 --reorder ... -m 3)
    r = subprocess.run([REF_BIN, "--wrapper", "basic-0", "-p", "1", "-S", "--sam-nohead"] +
                       args + ["-x", base] + inputs + [out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-400:]
    with open(out, "rb") as f:
        return f.read().split(b"\n")


@pytest.mark.parametrize("name,pol,args,length,n", [
    ("n2", dict(mode="n", mms=2, seed_len=28, qual_thresh=70), ["-n", "2", "-l", "28", "-e", "70"], 100, 40000),
    ("v2_k2", dict(mode="v", mms=2, khits=2), ["-v", "2", "-k", "2"], 76, 30000),
    ("n2_best_strata_m3", dict(mode="n", mms=2, best=True, strata=True, mhits=3, max_bts=800), ["-n", "2", "--best", "--strata", "-m", "3"], 50, 20000),
])
def test_gpu_sam_equals_reference_on_100mbp_index(name, pol, args, length, n, scale):
    import bench
    from bowtie_amd.synth import synth_reads, write_fastq
    b = synth_reads(scale["text"], n, length, mm_dist=(0, 1, 2, 2, 3, 4), seed=77 + length)
    fq = os.path.join(scale["dir"], name + ".fq")
    write_fastq(b, fq)
    want = _reference_sam(scale["base"], args, [fq], os.path.join(scale["dir"], name + ".sam"))
    from bowtie_amd import hostio as H
    pl = A.make_policy(**pol)
    al = AL.Aligner(scale["idx"], pl)
    cap = max(1, int(pol.get("khits", 1)), 3 if pol.get("strata") else 1)
    res = al.align(b, hit_cap=cap)
    hits, n_hits, status, pool = H.pack_hits(res, cap)
    text, _ = H.format_hits(b, hits, n_hits, status, pool, cap, scale["idx"].refnames, scale["idx"].reflens,
                            H.out_opts(sam=True, khits=int(pol.get("khits", 1)), mhits=int(pol.get("mhits", 0xFFFFFFFF))))
    got = text.split(b"\n")
    bad = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert len(got) == len(want) and not bad, (name, len(got), len(want), got[bad[0]] if bad else None, want[bad[0]] if bad else None)
    aligned = sum(1 for x in want if x and x.split(b"\t")[1] != b"4")
    assert aligned > 0.6 * n


def test_gpu_paired_sam_equals_reference_on_100mbp_index(scale):
    """BASELINE config 5's shape (-1/-2 -n 1 --best -X 250, 50-bp mates) at this size."""
    import bench
    from bowtie_amd.synth import synth_pairs, write_fastq
    n = 10000
    b1, b2 = synth_pairs(scale["text"], n, 50, mm_dist=(0, 0, 1, 1, 2), seed=555)
    f1, f2 = os.path.join(scale["dir"], "pe_1.fq"), os.path.join(scale["dir"], "pe_2.fq")
    write_fastq(b1, f1)
    write_fastq(b2, f2)
    want = _reference_sam(scale["base"], ["-n", "1", "-l", "28", "-e", "70", "--best", "-X", "250"], ["-1", f1, "-2", f2],
                          os.path.join(scale["dir"], "pe.sam"))
    pol = A.make_policy(mode="n", mms=1, seed_len=28, qual_thresh=70, best=True, max_ins=250, max_bts=800)
    got = bench._gpu_sam(scale["idx"], pol, [b1, b2]).split(b"\n")
    bad = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert len(got) == len(want) and not bad, (len(got), len(want), got[bad[0]] if bad else None, want[bad[0]] if bad else None)


# The library runs the best-first engine call by call on an index of this size (bt_api.cpp picks the loop by index size);
# BT_BEST_NESTED=0 puts the same inputs through the wavefront automaton (bt_best_kernel), against the live reference too.
def test_gpu_best_sam_equals_reference_on_100mbp_index_through_the_automaton(scale, monkeypatch):
    monkeypatch.setenv("BT_BEST_NESTED", "0")
    test_gpu_sam_equals_reference_on_100mbp_index(
        "n2_best_strata_m3_automaton", dict(mode="n", mms=2, best=True, strata=True, mhits=3, max_bts=800),
        ["-n", "2", "--best", "--strata", "-m", "3"], 50, 20000, scale)


def test_gpu_paired_sam_equals_reference_on_100mbp_index_through_the_automaton(scale, monkeypatch):
    monkeypatch.setenv("BT_BEST_NESTED", "0")
    test_gpu_paired_sam_equals_reference_on_100mbp_index(scale)
