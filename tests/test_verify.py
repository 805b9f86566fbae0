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
