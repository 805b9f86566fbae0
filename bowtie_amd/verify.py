"""Independent check of reported alignments against the reference text (torch, any device): a
size-independent property used by the tests and by `bench.py --verify` at full benchmark size.

It does not search: for every read with a hit it fetches the reference window the hit names,
compares it base by base with the read as aligned, and checks that
  * the mismatch list (positions from the 5' end, reference bases) is exactly the set of differing
    columns (Hit::mms / Hit::refcs, ebwt.h:1288-1405),
  * the alignment is admissible under the policy: `-v k`: at most k mismatches; `-n k -l L -e E`: at
    most k in the first L bases from the 5' end and a Maq-rounded quality sum of all mismatches <= E
    (qual.cpp:4-32, ebwt_search_backtrack.h:456-739),
  * stratum and cost say the same (stratum = seed mismatches, cost = quality sum | stratum << 14;
    ebwt_search_backtrack.h:1164-1200),
  * the window does not straddle a fragment boundary (joinedToTextOff, ebwt.h:2569-2629).
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np


def read_fragments(base: str):
    """(plen [nPat], rstarts [nFrag, 3] = joined start, reference index, offset in the reference) from
    the header of <base>.1.ebwt (ebwt.h:2926-3000)."""
    with open(base + ".1.ebwt", "rb") as f:
        one, ln, line_rate, lines_per_side, off_rate, ftab_chars, flags = struct.unpack("<iIiiiii", f.read(28))
        assert one == 1
        (npat,) = struct.unpack("<I", f.read(4))
        plen = np.frombuffer(f.read(4 * npat), dtype="<u4").copy()
        (nfrag,) = struct.unpack("<I", f.read(4))
        rstarts = np.frombuffer(f.read(12 * nfrag), dtype="<u4").reshape(nfrag, 3).copy()
    return ln, plen, rstarts


def verify_hits(text_t, text_len: int, rstarts: np.ndarray, seq, qual, length: int, hits_u8, n_hits, mm_pool,
                pol: Dict, chunk: int = 4_000_000, max_mm: int = 24) -> Dict[str, int]:
    """seq/qual [n, stride] u8, hits_u8 [n * 24] u8 (hit_cap 1), n_hits [n] i32, mm_pool i16/u16 --
    all torch tensors on text_t's device; equal-length reads.  -> counts of checked / failing hits per rule."""
    import torch
    dev = text_t.device
    n = seq.shape[0]
    L = length
    mode_n = pol.get("mode", "n") == "n"
    k_mm = int(pol.get("mms", 2))
    seed_len = int(pol.get("seed_len", 28))
    qthr = int(pol.get("qual_thresh", 70))
    maq = bool(pol.get("maq_round", True))
    rs = torch.from_numpy(rstarts.astype(np.int64)).to(dev)
    order = torch.argsort(rs[:, 1] * (1 << 32) + rs[:, 2])
    rs = rs[order]
    frag_key = rs[:, 1] * (1 << 32) + rs[:, 2]
    frag_end = torch.cat([rs[1:, 0], torch.tensor([text_len], device=dev)])      # joined end, valid in joined order only
    # joined ends: recompute in joined order
    by_joined = torch.argsort(rs[:, 0])
    ends = torch.empty_like(rs[:, 0])
    js = rs[by_joined, 0]
    ends[by_joined] = torch.cat([js[1:], torch.tensor([text_len], device=dev)])
    del frag_end
    H = hits_u8.view(torch.int32).view(-1, 6)
    pool = mm_pool.view(torch.int16).to(torch.int32) & 0xFFFF
    ar = torch.arange(L, device=dev)
    out = dict(checked=0, bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        sel = (n_hits[lo:hi] > 0).nonzero().flatten() + lo
        m = int(sel.numel())
        if m == 0:
            continue
        h = H[sel].to(torch.int64)
        tidx, toff = h[:, 0] & 0xFFFFFFFF, h[:, 1] & 0xFFFFFFFF
        mm_off = h[:, 3] & 0xFFFFFFFF
        cost, nmm = h[:, 4] & 0xFFFF, (h[:, 4] >> 16) & 0xFFFF
        stratum, fw = h[:, 5] & 0xFF, ((h[:, 5] >> 8) & 0xFF) != 0
        f = torch.searchsorted(frag_key, tidx * (1 << 32) + toff, right=True) - 1
        f = f.clamp(min=0)
        joined = rs[f, 0] + (toff - rs[f, 2])
        bad_window = (rs[f, 1] != tidx) | (toff < rs[f, 2]) | (joined + L > ends[f])
        joined = joined.clamp(0, text_len - L)
        win = text_t[joined[:, None] + ar[None, :]]
        rd = seq[sel][:, :L]
        ql = qual[sel][:, :L].to(torch.int64) - 33
        rc = torch.where(rd < 4, 3 - rd, rd).flip(1)
        ori = torch.where(fw[:, None], rd, rc)
        qori = torch.where(fw[:, None], ql, ql.flip(1))
        mism = win != ori
        cnt = mism.sum(1)
        pos5 = torch.where(fw[:, None], ar[None, :], (L - 1 - ar)[None, :])
        pen = qori if not maq else torch.where(qori < 5, 0, torch.where(qori < 15, 10, torch.where(qori < 25, 20, 30)))
        qsum = (pen * mism).sum(1)
        seedmm = (mism & (pos5 < seed_len)).sum(1)
        if mode_n:
            bad_policy = (seedmm > k_mm) | (qsum > qthr)
            bad_cost = (stratum != seedmm) | ((cost & 0x3FFF) != qsum) | ((cost >> 14) != stratum)
        else:
            bad_policy = cnt > k_mm
            bad_cost = (stratum != cnt) | ((cost >> 14) != stratum)
        # the mismatch list: every entry names a differing column with the reference base there
        listed = torch.zeros_like(mism)
        bad_list = torch.zeros(m, dtype=torch.bool, device=dev)
        rows = torch.arange(m, device=dev)
        for k in range(max_mm):
            has = nmm > k
            if not bool(has.any()):
                break
            e = pool[(mm_off + k).clamp(max=pool.numel() - 1)]
            p5 = (e & 0x3FF).to(torch.int64)
            refc = ((e >> 12) & 3).to(torch.uint8)
            col = torch.where(fw, p5, L - 1 - p5).clamp(0, L - 1)
            ok = (p5 < L) & mism[rows, col] & (win[rows, col] == refc) & ~listed[rows, col]
            bad_list |= has & ~ok
            listed[rows[has], col[has]] = True
        bad_list |= (nmm > max_mm)
        out["checked"] += m
        out["bad_window"] += int(bad_window.sum())
        out["bad_mm_count"] += int((cnt != nmm).sum())
        out["bad_mm_list"] += int(bad_list.sum())
        out["bad_policy"] += int(bad_policy.sum())
        out["bad_cost"] += int(bad_cost.sum())
    return out
