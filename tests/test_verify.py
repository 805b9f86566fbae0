"""bowtie_amd/verify.py -- the text-based re-check of reported alignments used at full benchmark size
(`bench.py --verify`) -- must accept what the oracle reports and reject tampered hits."""
import os

import numpy as np
import pytest
import torch

import common as T
from bowtie_amd import hostio as H
from bowtie_amd import verify as V


def tensors(index, batch, per):
    hits, nh, st, pool = H.pack_hits(per, 1)
    ln, plen, rstarts = V.read_fragments(os.path.join(T.G, index))
    text = torch.from_numpy(T.joined_text(index).copy())
    return dict(text_t=text, text_len=ln, rstarts=rstarts, seq=torch.from_numpy(batch.seq.copy()),
                qual=torch.from_numpy(batch.qual.copy()), hits_u8=torch.from_numpy(hits.view(np.uint8).copy()),
                n_hits=torch.from_numpy(nh.astype(np.int32)), mm_pool=torch.from_numpy(pool.view(np.int16).copy()))


@pytest.mark.parametrize("index,reads,mode,length", [("e_coli", "syn100", "n2", 100), ("multi", "syn100", "n2", 100),
                                                     ("multi", "syn76", "v2", 76), ("e_coli", "syn36", "v0", 36),
                                                     ("multi", "syn100", "n3", 100), ("multi", "syn76", "n2_nomaq", 76),
                                                     ("e_coli", "syn100", "n2_l20_e100", 100)])
def test_oracle_hits_verify_clean(index, reads, mode, length):
    batch = T.read_set(index, reads)
    kw = T.MODES[mode]
    per = T.oracle_results(index, batch, kw, cap=1)
    t = tensors(index, batch, per)
    r = V.verify_hits(length=length, pol=kw, chunk=97, **t)
    assert r["checked"] == sum(1 for h, _, _ in per if h) > 50
    assert {k: v for k, v in r.items() if k != "checked"} == dict(bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0)


def test_tampered_hits_are_caught():
    batch = T.read_set("multi", "syn100")
    kw = T.MODES["n2"]
    per = T.oracle_results("multi", batch, kw, cap=1)
    t = tensors("multi", batch, per)
    H6 = t["hits_u8"].view(torch.int32).view(-1, 6)
    aligned = (t["n_hits"] > 0).nonzero().flatten()
    H6[aligned[0], 1] += 1                      # shifted offset
    H6[aligned[1], 5] ^= 0x100                  # wrong strand
    i2 = [int(i) for i in aligned[4:] if (int(H6[i, 4]) >> 16) > 0][0]
    t["mm_pool"][int(H6[i2, 3])] ^= 0x1000      # wrong reference base in a mismatch entry
    r = V.verify_hits(length=100, pol=kw, **t)
    assert r["bad_mm_count"] >= 2 and r["bad_mm_list"] >= 1
    # an offset that runs over a fragment end
    ln, plen, rstarts = V.read_fragments(os.path.join(T.G, "multi"))
    H6[aligned[3], 0] = int(rstarts[0, 1]); H6[aligned[3], 1] = int(rstarts[0, 2]) + int(rstarts[1, 0]) - 50
    r = V.verify_hits(length=100, pol=kw, **t)
    assert r["bad_window"] >= 1


def pair_tensors(index, b1, b2, per, cap=2):
    hits, nh, st, pool = H.pack_hits(per, cap)
    ln, plen, rstarts = V.read_fragments(os.path.join(T.G, index))
    text = torch.from_numpy(T.joined_text(index).copy())
    return dict(text_t=text, text_len=ln, rstarts=rstarts, seq1=torch.from_numpy(b1.seq.copy()), qual1=torch.from_numpy(b1.qual.copy()),
                seq2=torch.from_numpy(b2.seq.copy()), qual2=torch.from_numpy(b2.qual.copy()),
                hits_u8=torch.from_numpy(hits.view(np.uint8).copy()), n_hits=torch.from_numpy(nh.astype(np.int32)),
                mm_pool=torch.from_numpy(pool.view(np.int16).copy()))


def revcomp_batch(b):
    import copy
    r = copy.copy(b)
    r.seq, r.qual = b.seq.copy(), b.qual.copy()
    for i in range(b.n):
        L = int(b.len[i])
        s = b.seq[i, :L][::-1]
        r.seq[i, :L] = np.where(s < 4, 3 - s, s)
        r.qual[i, :L] = b.qual[i, :L][::-1]
    return r


PAIR_RULES = dict(bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0, bad_mates=0, bad_order=0,
                  bad_orientation=0, bad_insert=0, bad_containment=0, odd_count=0)


@pytest.mark.parametrize("index,reads,mode,v1", [
    ("e_coli", "pe50", "pe_n1_best_X500", False), ("e_coli", "pe50", "pe_n2_best_X500_ff", False),
    ("e_coli", "pe50", "pe_n2_best_X500_rf", False), ("e_coli", "pe50", "pe_n2_best_X400_I250_k3", False),
    ("e_coli", "pe50", "pe_v2_best_X500", False), ("multi", "pe50", "pe_n2_best_X500", False),
    ("e_coli", "pe50", "pe_n2_best_X500", True)])
def test_oracle_pairs_verify_clean(index, reads, mode, v1):
    """what the oracle reports for pairs (the golden-pinned PairedBWAlignerV2 / V1 restatements) passes every rule of
    verify_pairs -- with --ff / --rf, -I and -k 3 (the first pair of each is the one re-derived)"""
    b1, b2 = T.pair_set(index, reads)
    # the synthetic pairs are --fr fragments: --ff wants mate 2 as its reverse complement, --rf both mates
    if mode.endswith("_ff"):
        b2 = revcomp_batch(b2)
    elif mode.endswith("_rf"):
        b1, b2 = revcomp_batch(b1), revcomp_batch(b2)
    per = T.oracle_pair_results(index, b1, b2, T.MODES[mode], v1=v1)
    cap = max(2, max(len(h) for h, _, _ in per))
    t = pair_tensors(index, b1, b2, per, cap=cap)
    r = V.verify_pairs(len1=50, len2=50, pol=T.MODES[mode], hit_cap=cap, chunk=61, **t)
    assert r["checked"] == sum(1 for h, _, _ in per if len(h) >= 2) > 50
    assert {k: v for k, v in r.items() if k != "checked"} == PAIR_RULES


def test_tampered_pairs_are_caught():
    b1, b2 = T.pair_set("e_coli", "pe50")
    mode = T.MODES["pe_n1_best_X500"]
    per = T.oracle_pair_results("e_coli", b1, b2, mode)
    t = pair_tensors("e_coli", b1, b2, per)
    H6 = t["hits_u8"].view(torch.int32).view(-1, 2, 6)
    al = (t["n_hits"] >= 2).nonzero().flatten()
    base = V.verify_pairs(len1=50, len2=50, pol=mode, **t)
    assert {k: v for k, v in base.items() if k != "checked"} == PAIR_RULES

    def with_change(fn):
        keep = H6.clone()
        fn()
        r = V.verify_pairs(len1=50, len2=50, pol=mode, **t)
        H6.copy_(keep)
        return r
    i = int(al[0])
    r = with_change(lambda: H6[i].copy_(H6[i].flip(0)))                                   # downstream mate first
    assert r["bad_order"] == 1
    r = with_change(lambda: H6[i, 1, 5].copy_(H6[i, 1, 5] ^ 0x30000))                     # both records say the same mate
    assert r["bad_mates"] == 1
    r = with_change(lambda: H6[i, 1, 1].add_(600))                                        # fragment longer than -X
    assert r["bad_insert"] == 1 and r["bad_mm_count"] >= 1
    r = with_change(lambda: (H6[i, 0, 5].copy_(H6[i, 0, 5] ^ 0x100), H6[i, 1, 5].copy_(H6[i, 1, 5] ^ 0x100)))
    assert r["bad_orientation"] == 1                                                       # both strands flipped: not --fr
    r = with_change(lambda: H6[i, 1, 1].copy_(H6[i, 0, 1]))                                # same start: one contains the other
    assert r["bad_containment"] == 1
    r = V.verify_pairs(len1=50, len2=50, pol=dict(mode, max_ins=60), **t)                  # the policy, not the hits
    assert r["bad_insert"] > 10
    t["n_hits"][int(al[1])] = 3
    assert V.verify_pairs(len1=50, len2=50, pol=mode, **t)["odd_count"] == 1


@pytest.mark.parametrize("l1,l2,extra", [(50, 36, {}), (36, 50, {}), (50, 36, {"allow_contain": True}), (40, 40, {"min_ins": 200, "max_ins": 420})])
def test_oracle_pairs_of_unequal_mates_verify_clean(l1, l2, extra):
    """mates of different lengths (the containment rule depends on which mate is the shorter one), --allow-contain, and
    -I / -X that actually cut: the oracle's pairs pass every rule"""
    import copy
    b1, b2 = T.pair_set("e_coli", "pe50")

    def trimmed(b, L):
        r = copy.copy(b)
        r.seq, r.qual, r.len = b.seq.copy(), b.qual.copy(), np.minimum(b.len, L).astype(b.len.dtype)
        r.seq[:, L:] = 4
        return r
    b1, b2 = trimmed(b1, l1), trimmed(b2, l2)
    mode = dict(T.MODES["pe_n2_best_X500"], **extra)
    per = T.oracle_pair_results("e_coli", b1, b2, mode)
    t = pair_tensors("e_coli", b1, b2, per)
    r = V.verify_pairs(len1=l1, len2=l2, pol=mode, **t)
    assert r["checked"] == sum(1 for h, _, _ in per if len(h) >= 2) > 40
    assert {k: v for k, v in r.items() if k != "checked"} == PAIR_RULES
