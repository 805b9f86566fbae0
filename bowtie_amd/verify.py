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
`verify_pairs` does the same for both mates of a reported pair and re-derives what makes them a pair (same reference,
upstream mate first, orientation, fragment length, containment).
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


class _Frags:
    """The index's fragment table (rstarts) in both orders, on the text's device."""

    def __init__(self, rstarts: np.ndarray, text_len: int, dev):
        import torch
        rs = torch.from_numpy(rstarts.astype(np.int64)).to(dev)
        order = torch.argsort(rs[:, 1] * (1 << 32) + rs[:, 2])
        self.rs = rs = rs[order]
        self.key = rs[:, 1] * (1 << 32) + rs[:, 2]
        by_joined = torch.argsort(rs[:, 0])
        self.ends = torch.empty_like(rs[:, 0])                    # joined end of each fragment
        js = rs[by_joined, 0]
        self.ends[by_joined] = torch.cat([js[1:], torch.tensor([text_len], device=dev)])
        self.text_len = text_len


def _policy(pol: Dict):
    return (pol.get("mode", "n") == "n", int(pol.get("mms", 2)), int(pol.get("seed_len", 28)),
            int(pol.get("qual_thresh", 70)), bool(pol.get("maq_round", True)))


def _check_records(text_t, fr: "_Frags", h, rd, ql, L: int, pol: Dict, pool, max_mm: int, ham_may_be_zero: bool = False):
    """h [m, 6] int64 (bt_hit as six dwords), rd / ql [m, L] the reads as given (5' to 3', codes / Phred): the per-record
    verdicts (bool [m] each) of the four rules of the module's header."""
    import torch
    dev = text_t.device
    mode_n, k_mm, seed_len, qthr, maq = _policy(pol)
    m = h.shape[0]
    rs, ends = fr.rs, fr.ends
    ar = torch.arange(L, device=dev)
    tidx, toff = h[:, 0] & 0xFFFFFFFF, h[:, 1] & 0xFFFFFFFF
    mm_off = h[:, 3] & 0xFFFFFFFF
    cost, nmm = h[:, 4] & 0xFFFF, (h[:, 4] >> 16) & 0xFFFF
    stratum, fw = h[:, 5] & 0xFF, ((h[:, 5] >> 8) & 0xFF) != 0
    f = torch.searchsorted(fr.key, tidx * (1 << 32) + toff, right=True) - 1
    f = f.clamp(min=0)
    joined = rs[f, 0] + (toff - rs[f, 2])
    bad_window = (rs[f, 1] != tidx) | (toff < rs[f, 2]) | (joined + L > ends[f])
    joined = joined.clamp(0, fr.text_len - L)
    win = text_t[joined[:, None] + ar[None, :]]
    ql = ql.to(torch.int64) - 33
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
        ham = cost & 0x3FFF
        # a pair's records: the mate found through the index (the anchor) carries the range's cost as the search left it,
        # whose quality part the reference leaves at 0 there; the mate found in the reference window carries its quality sum
        # (ref_aligner.h:63-101).  Neither is shown to the user; the policy is re-derived from the text either way.
        bad_ham = ((ham != qsum) & (ham != 0)) if ham_may_be_zero else (ham != qsum)
        bad_cost = (stratum != seedmm) | bad_ham | ((cost >> 14) != stratum)
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
    return dict(bad_window=bad_window, bad_mm_count=cnt != nmm, bad_mm_list=bad_list, bad_policy=bad_policy, bad_cost=bad_cost)


def verify_hits(text_t, text_len: int, rstarts: np.ndarray, seq, qual, length: int, hits_u8, n_hits, mm_pool,
                pol: Dict, chunk: int = 4_000_000, max_mm: int = 24) -> Dict[str, int]:
    """seq/qual [n, stride] u8, hits_u8 [n * 24] u8 (hit_cap 1), n_hits [n] i32, mm_pool i16/u16 --
    all torch tensors on text_t's device; equal-length reads.  -> counts of checked / failing hits per rule."""
    import torch
    dev = text_t.device
    n = seq.shape[0]
    L = length
    fr = _Frags(rstarts, text_len, dev)
    H = hits_u8.view(torch.int32).view(-1, 6)
    pool = mm_pool.view(torch.int16).to(torch.int32) & 0xFFFF
    out = dict(checked=0, bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        sel = (n_hits[lo:hi] > 0).nonzero().flatten() + lo
        m = int(sel.numel())
        if m == 0:
            continue
        v = _check_records(text_t, fr, H[sel].to(torch.int64), seq[sel][:, :L], qual[sel][:, :L], L, pol, pool, max_mm)
        out["checked"] += m
        for k, b in v.items():
            out[k] += int(b.sum())
    return out


def verify_pairs(text_t, text_len: int, rstarts: np.ndarray, seq1, qual1, seq2, qual2, len1: int, len2: int, hits_u8, n_hits,
                 mm_pool, pol: Dict, hit_cap: int = 2, chunk: int = 2_000_000, max_mm: int = 24) -> Dict[str, int]:
    """The same re-check for paired results (bt_align_pairs' layout: hit_cap slots per pair, a pair's two records adjacent,
    upstream mate first, bt_hit.pad[0] = 1 / 2; n_hits counts mate alignments).  The first reported pair of every pair of
    reads that has one is re-derived: both mates' windows, mismatch lists, policy and cost as for single reads, and what makes
    the two a pair (PairedBWAlignerV2::resolveOutstandingInRef / report, aligner.h:1863-1990; V1's rules are the same,
    aligner.h:1283-1420):
      * one record of each mate, on the same reference sequence, the upstream (leftmost) one first;
      * orientation (ebwt_search.cpp:902-906): either mate 1 is upstream with strand `mate1_fw` and mate 2 downstream with
        strand `mate2_fw`, or the whole fragment is reverse-complemented: mate 2 upstream with strand !mate2_fw, mate 1
        downstream with strand !mate1_fw;
      * fragment length (upstream mate's first base to downstream mate's last) within [-I, -X], and -X longer than either mate;
      * unless --allow-contain: neither mate's window contains the other's (aligner.h:1934-1968 -- for the anchor upstream:
        downstream start > upstream start, and its end beyond the upstream mate's end if it is the shorter one; for the
        anchor downstream: the mirror image; a pair passes if either reading admits it, as the re-check cannot know which
        mate was the anchor).
    Equal-length mates (len1, len2).  -> counts per rule, `checked` = pairs looked at."""
    import torch
    dev = text_t.device
    n = seq1.shape[0]
    fr = _Frags(rstarts, text_len, dev)
    H = hits_u8.view(torch.int32).view(-1, hit_cap, 6)
    pool = mm_pool.view(torch.int16).to(torch.int32) & 0xFFFF
    fw1, fw2 = bool(pol.get("mate1_fw", True)), bool(pol.get("mate2_fw", False))
    minins, maxins = int(pol.get("min_ins", 0)), int(pol.get("max_ins", 250))
    contain = bool(pol.get("allow_contain", False))
    out = dict(checked=0, bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0,
               bad_mates=0, bad_order=0, bad_orientation=0, bad_insert=0, bad_containment=0, odd_count=0)
    out["odd_count"] = int(((n_hits.to(torch.int64) & 1) != 0).sum())
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        sel = (n_hits[lo:hi] >= 2).nonzero().flatten() + lo
        m = int(sel.numel())
        if m == 0:
            continue
        hu, hd = H[sel, 0].to(torch.int64), H[sel, 1].to(torch.int64)          # upstream, downstream record
        mu, md = (hu[:, 5] >> 16) & 0xFF, (hd[:, 5] >> 16) & 0xFF
        bad_mates = ~(((mu == 1) & (md == 2)) | ((mu == 2) & (md == 1)))
        up_is_1 = mu == 1
        for which, hrec, is1 in (("u", hu, up_is_1), ("d", hd, ~up_is_1)):
            # a record's read is mate 1 or mate 2 according to its own flag: check each group with its own read rows
            for mate, seq, qual, L in ((1, seq1, qual1, len1), (2, seq2, qual2, len2)):
                g = (is1 if mate == 1 else ~is1).nonzero().flatten()
                if g.numel() == 0:
                    continue
                v = _check_records(text_t, fr, hrec[g], seq[sel[g]][:, :L], qual[sel[g]][:, :L], L, pol, pool, max_mm,
                                   ham_may_be_zero=True)
                for k, b in v.items():
                    out[k] += int(b.sum())
        lu = torch.where(up_is_1, len1, len2)
        ld = torch.where(up_is_1, len2, len1)
        tu, td = hu[:, 0] & 0xFFFFFFFF, hd[:, 0] & 0xFFFFFFFF
        pu, pd = hu[:, 1] & 0xFFFFFFFF, hd[:, 1] & 0xFFFFFFFF
        fu, fd = ((hu[:, 5] >> 8) & 0xFF) != 0, ((hd[:, 5] >> 8) & 0xFF) != 0
        bad_order = (tu != td) | (pu > pd)
        want_fu = torch.where(up_is_1, torch.tensor(fw1, device=dev), torch.tensor(not fw2, device=dev))
        want_fd = torch.where(up_is_1, torch.tensor(fw2, device=dev), torch.tensor(not fw1, device=dev))
        bad_orient = (fu != want_fu) | (fd != want_fd)
        frag = pd + ld - pu
        bad_insert = (frag < minins) | (frag > maxins) | (maxins <= max(len1, len2))
        if contain:
            bad_contain = torch.zeros(m, dtype=torch.bool, device=dev)
        else:
            # anchor upstream (alen = lu, qlen = ld): pd >= pu + 1, + (lu - ld) more if ld < lu
            ok_a = pd >= pu + 1 + (lu - ld).clamp(min=0)
            # anchor downstream (alen = ld, qlen = lu): pu + lu <= pd + min(ld, lu) - 1
            ok_b = pu + lu <= pd + torch.minimum(ld, lu) - 1
            bad_contain = ~(ok_a | ok_b)
        out["checked"] += m
        out["bad_mates"] += int(bad_mates.sum())
        out["bad_order"] += int(bad_order.sum())
        out["bad_orientation"] += int(bad_orient.sum())
        out["bad_insert"] += int(bad_insert.sum())
        out["bad_containment"] += int(bad_contain.sum())
    return out
