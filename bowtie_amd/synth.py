"""Seeded synthetic reads: sampled uniformly from a joined reference text, strand
50/50, a fixed small distribution of planted substitutions, Phred uniform 10..40, a fraction of
reads with one N, names r<i>.  numpy here (tests, small benches); bench.py has the torch/HBM twin
(`synth_reads_torch`) that produces the same kind of batch directly on the GPU."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .reads import ReadBatch, rand_seeds

_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def synth_reads(text: np.ndarray, n: int, length: int, mm_dist: Sequence[int] = (0, 1, 2, 2, 3, 4),
                seed: int = 12345, n_frac: float = 0.01, qlo: int = 10, qhi: int = 40,
                lowq_frac: float = 0.0, global_seed: int = 0, stride: int | None = None,
                name_prefix: str = "r") -> ReadBatch:
    """text: uint8 codes 0..3 of the joined reference."""
    rng = np.random.default_rng(seed)
    T = len(text)
    start = rng.integers(0, T - length + 1, size=n)
    idx = start[:, None] + np.arange(length)[None, :]
    seq = text[idx].astype(np.uint8)
    rc = rng.random(n) < 0.5
    seq[rc] = _COMP[seq[rc][:, ::-1]]
    # planted substitutions
    nmm = np.asarray(mm_dist)[rng.integers(0, len(mm_dist), size=n)]
    for k in range(int(max(mm_dist)) if len(mm_dist) else 0):
        rows = np.nonzero(nmm > k)[0]
        pos = rng.integers(0, length, size=len(rows))
        seq[rows, pos] = (seq[rows, pos] + rng.integers(1, 4, size=len(rows))) & 3
    # Ns
    if n_frac > 0:
        rows = np.nonzero(rng.random(n) < n_frac)[0]
        seq[rows, rng.integers(0, length, size=len(rows))] = 4
    qual = rng.integers(qlo, qhi + 1, size=(n, length)).astype(np.uint8)
    if lowq_frac > 0:
        low = rng.random((n, length)) < lowq_frac
        qual[low] = rng.integers(0, 5, size=int(low.sum()))
    qual = (qual + 33).astype(np.uint8)
    stride = stride or max(16, (length + 15) & ~15)
    pseq = np.full((n, stride), 4, dtype=np.uint8)
    pqual = np.full((n, stride), 33, dtype=np.uint8)
    pseq[:, :length] = seq
    pqual[:, :length] = qual
    lens = np.full(n, length, dtype=np.uint16)
    names = [("%s%d" % (name_prefix, i)).encode() for i in range(n)]
    seeds = rand_seeds(pseq, pqual, lens, names, global_seed)
    return ReadBatch(pseq, pqual, lens, seeds, names)


def write_fastq(batch: ReadBatch, path: str) -> None:
    asc = np.frombuffer(b"ACGTN", dtype=np.uint8)
    L = int(batch.len[0]) if batch.n else 0
    if batch.n and (batch.len == L).all() and all(nm == b"r%d" % i for i, nm in zip(range(3), batch.names[:3])):
        # fixed-length fast path (bench samples): assemble the records as one byte matrix
        n = batch.n
        names = np.char.add("@r", np.arange(n).astype(str)).astype("S")
        w = names.dtype.itemsize
        rec = np.full((n, w + 1 + L + 3 + L + 1), ord("\n"), dtype=np.uint8)
        nm = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, w)
        rec[:, :w] = np.where(nm == 0, ord(" "), nm)            # pad short names (stripped below)
        rec[:, w + 1:w + 1 + L] = asc[batch.seq[:, :L]]
        rec[:, w + 2 + L] = ord("+")
        rec[:, w + 4 + L:w + 4 + 2 * L] = batch.qual[:, :L]
        data = rec.tobytes()
        if w > len(b"@r0"):
            import re
            data = re.sub(rb"(@r\d+) +\n", rb"\1\n", data)
        with open(path, "wb") as f:
            f.write(data)
        return
    with open(path, "wb") as f:
        for i in range(batch.n):
            L = int(batch.len[i])
            f.write(b"@" + batch.names[i] + b"\n" + asc[batch.seq[i, :L]].tobytes() + b"\n+\n" +
                    batch.qual[i, :L].tobytes() + b"\n")


# ---- torch twin: batches generated directly in HBM (bench.py) -------------------------------
def synth_reads_torch(text_t, n: int, length: int, mm_dist=(0, 1, 2, 2, 3, 4), seed: int = 12345,
                      n_frac: float = 0.01, qlo: int = 10, qhi: int = 40, first_id: int = 0,
                      global_seed: int = 0, stride: int | None = None, chunk: int = 2_000_000,
                      shard: tuple | None = None):
    """text_t: uint8 torch tensor (codes 0..3) on the target device.  Returns a dict of device
    tensors {seq [n,stride] u8, qual [n,stride] u8, len [n] i16 (bit pattern of u16), seed [n] i32
    (bit pattern of u32)} laid out as bt_read_batch wants.  Same distribution as synth_reads()
    (not the same stream: torch's generator); per-read seeds follow genRandSeed (pat.cpp:21-57)
    with names r<first_id + i>.
    shard = (lo, hi): only reads [lo, hi) of the n-read set (every chunk of `chunk` reads has its own
    generator stream, so a shard holds exactly the reads the whole set has at those positions)."""
    import torch
    dev = text_t.device
    if shard is not None:
        lo_s, hi_s = shard
        parts = []
        for c0 in range((lo_s // chunk) * chunk, hi_s, chunk):
            m = min(chunk, n - c0)
            part = synth_reads_torch(text_t, m, length, mm_dist, seed * 1000003 + c0 // chunk + 1, n_frac, qlo, qhi,
                                     first_id + c0, global_seed, stride, chunk)
            a, b = max(lo_s, c0) - c0, min(hi_s, c0 + m) - c0
            parts.append({k: (v[a:b] if hasattr(v, "shape") else v) for k, v in part.items()})
        out = {k: (torch.cat([p_[k] for p_ in parts]) if hasattr(parts[0][k], "shape") else parts[0][k]) for k in parts[0]}
        out["n"] = hi_s - lo_s
        return out
    stride = stride or max(16, (length + 15) & ~15)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    T = text_t.numel()
    seq = torch.full((n, stride), 4, dtype=torch.uint8, device=dev)
    qual = torch.full((n, stride), 33, dtype=torch.uint8, device=dev)
    seeds = torch.empty(n, dtype=torch.int32, device=dev)
    ar = torch.arange(length, device=dev)
    mmd = torch.tensor(list(mm_dist), device=dev)
    base = ((global_seed + 101) * 59 * 61 * 67 * 71 * 73 * 79 * 83) & 0xFFFFFFFF
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        start = torch.randint(0, T - length + 1, (m,), generator=g, device=dev)
        s = text_t[(start[:, None] + ar[None, :])]
        rc = torch.rand(m, generator=g, device=dev) < 0.5
        s = torch.where(rc[:, None], 3 - s.flip(1), s)
        nmm = mmd[torch.randint(0, len(mm_dist), (m,), generator=g, device=dev)]
        rows = torch.arange(m, device=dev)
        for k in range(int(max(mm_dist)) if len(mm_dist) else 0):
            pos = torch.randint(0, length, (m,), generator=g, device=dev)
            add = torch.randint(1, 4, (m,), generator=g, device=dev).to(torch.uint8)
            cur = s[rows, pos]
            s[rows, pos] = torch.where(nmm > k, (cur + add) & 3, cur)
        if n_frac > 0:
            hasn = torch.rand(m, generator=g, device=dev) < n_frac
            pos = torch.randint(0, length, (m,), generator=g, device=dev)
            cur = s[rows, pos]
            s[rows, pos] = torch.where(hasn, torch.full_like(cur, 4), cur)
        q = (torch.randint(qlo, qhi + 1, (m, length), generator=g, device=dev) + 33).to(torch.uint8)
        seq[lo:lo + m, :length] = s
        qual[lo:lo + m, :length] = q
        # genRandSeed
        acc = torch.full((m,), base, dtype=torch.int64, device=dev)
        for i in range(length):
            acc ^= s[:, i].to(torch.int64) << ((i & 15) << 1)
            acc ^= q[:, i].to(torch.int64) << ((i & 3) << 3)
        ids = torch.arange(first_id + lo, first_id + lo + m, dtype=torch.int64, device=dev)
        acc ^= ord("r")
        ndig = torch.ones_like(ids)
        p = 10
        for _ in range(11):
            ndig += (ids >= p).to(torch.int64)
            p *= 10
        for k in range(12):                       # digit k from the left sits at name index k+1
            e = ndig - 1 - k
            valid = e >= 0
            div = torch.pow(torch.tensor(10, dtype=torch.int64, device=dev), e.clamp(min=0))
            dig = (ids // div) % 10 + ord("0")
            acc ^= torch.where(valid, dig << (((k + 1) & 3) << 3), torch.zeros_like(dig))
        acc &= 0xFFFFFFFF
        seeds[lo:lo + m] = torch.where(acc >= 2 ** 31, acc - 2 ** 32, acc).to(torch.int32)
    lens = torch.full((n,), length, dtype=torch.int16, device=dev)
    return {"seq": seq, "qual": qual, "len": lens, "seed": seeds, "n": n, "stride": stride, "length": length}


def synth_pairs(text: np.ndarray, n: int, length: int, frag_lo: int = 200, frag_hi: int = 450,
                mm_dist: Sequence[int] = (0, 0, 1, 1, 2), seed: int = 777, n_frac: float = 0.01,
                qlo: int = 10, qhi: int = 40, length2: int | None = None, name_prefix: str = "p"):
    """--fr read pairs: a fragment of frag_lo..frag_hi bases cut from the joined text on either
    strand; mate 1 = its first `length` bases, mate 2 = the reverse complement of its last
    `length2` bases.  Names <prefix><i>/1 and /2.  Returns (ReadBatch, ReadBatch)."""
    rng = np.random.default_rng(seed)
    length2 = length2 or length
    T = len(text)
    flen = rng.integers(max(frag_lo, length, length2), frag_hi + 1, size=n)
    start = rng.integers(0, T - frag_hi, size=n)
    out = []
    strand = rng.random(n) < 0.5
    for mate, L in ((1, length), (2, length2)):
        seq = np.empty((n, L), dtype=np.uint8)
        for i in range(n):
            frag = text[start[i]:start[i] + flen[i]]
            if strand[i]:
                frag = _COMP[frag[::-1]]
            seq[i] = frag[:L] if mate == 1 else _COMP[frag[-L:][::-1]]
        nmm = np.asarray(mm_dist)[rng.integers(0, len(mm_dist), size=n)]
        for k in range(int(max(mm_dist)) if len(mm_dist) else 0):
            rows = np.nonzero(nmm > k)[0]
            pos = rng.integers(0, L, size=len(rows))
            seq[rows, pos] = (seq[rows, pos] + rng.integers(1, 4, size=len(rows))) & 3
        if n_frac > 0:
            rows = np.nonzero(rng.random(n) < n_frac)[0]
            seq[rows, rng.integers(0, L, size=len(rows))] = 4
        qual = (rng.integers(qlo, qhi + 1, size=(n, L)) + 33).astype(np.uint8)
        stride = max(16, (L + 15) & ~15)
        pseq = np.full((n, stride), 4, dtype=np.uint8)
        pqual = np.full((n, stride), 33, dtype=np.uint8)
        pseq[:, :L] = seq
        pqual[:, :L] = qual
        lens = np.full(n, L, dtype=np.uint16)
        names = [("%s%d/%d" % (name_prefix, i, mate)).encode() for i in range(n)]
        out.append(ReadBatch(pseq, pqual, lens, rand_seeds(pseq, pqual, lens, names, 0), names))
    return out[0], out[1]


def synth_pairs_torch(text_t, n: int, length: int, frag_lo: int = 200, frag_hi: int = 450, mm_dist=(0, 0, 1, 1, 2),
                      seed: int = 777, n_frac: float = 0.01, qlo: int = 10, qhi: int = 40, first_id: int = 0,
                      chunk: int = 2_000_000):
    """Device twin of synth_pairs(): two dicts like synth_reads_torch()'s (mates 1 and 2), --fr pairs cut
    from fragments of frag_lo..frag_hi bases; names r<id>/1 and r<id>/2 (seeds follow genRandSeed over them)."""
    import torch
    dev = text_t.device
    stride = max(16, (length + 15) & ~15)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    T = text_t.numel()
    mmd = torch.tensor(list(mm_dist), device=dev)
    base = ((0 + 101) * 59 * 61 * 67 * 71 * 73 * 79 * 83) & 0xFFFFFFFF
    out = []
    for mate in (1, 2):
        out.append({"seq": torch.full((n, stride), 4, dtype=torch.uint8, device=dev),
                    "qual": torch.full((n, stride), 33, dtype=torch.uint8, device=dev),
                    "seed": torch.empty(n, dtype=torch.int32, device=dev),
                    "len": torch.full((n,), length, dtype=torch.int16, device=dev),
                    "n": n, "stride": stride, "length": length})
    ar = torch.arange(length, device=dev)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        flen = torch.randint(max(frag_lo, length), frag_hi + 1, (m,), generator=g, device=dev)
        start = torch.randint(0, T - frag_hi, (m,), generator=g, device=dev)
        strand = torch.rand(m, generator=g, device=dev) < 0.5
        rows = torch.arange(m, device=dev)
        ids = torch.arange(first_id + lo, first_id + lo + m, dtype=torch.int64, device=dev)
        ndig = torch.ones_like(ids)
        p = 10
        for _ in range(11):
            ndig += (ids >= p).to(torch.int64)
            p *= 10
        for mate in (1, 2):
            # the fragment on its strand: F = text[start:start+flen] or its reverse complement;
            # mate 1 = F[:L], mate 2 = revcomp(F[-L:])
            left = text_t[(start[:, None] + ar[None, :])]                               # first L bases of the text fragment
            right = text_t[((start + flen - length)[:, None] + ar[None, :])]            # last L bases
            if mate == 1:
                s_ = torch.where(strand[:, None], 3 - right.flip(1), left)
            else:
                s_ = torch.where(strand[:, None], left, 3 - right.flip(1))
            s_ = s_.clone()
            nmm = mmd[torch.randint(0, len(mm_dist), (m,), generator=g, device=dev)]
            for k in range(int(max(mm_dist)) if len(mm_dist) else 0):
                pos = torch.randint(0, length, (m,), generator=g, device=dev)
                add = torch.randint(1, 4, (m,), generator=g, device=dev).to(torch.uint8)
                cur = s_[rows, pos]
                s_[rows, pos] = torch.where(nmm > k, (cur + add) & 3, cur)
            if n_frac > 0:
                hasn = torch.rand(m, generator=g, device=dev) < n_frac
                pos = torch.randint(0, length, (m,), generator=g, device=dev)
                cur = s_[rows, pos]
                s_[rows, pos] = torch.where(hasn, torch.full_like(cur, 4), cur)
            q = (torch.randint(qlo, qhi + 1, (m, length), generator=g, device=dev) + 33).to(torch.uint8)
            o = out[mate - 1]
            o["seq"][lo:lo + m, :length] = s_
            o["qual"][lo:lo + m, :length] = q
            acc = torch.full((m,), base, dtype=torch.int64, device=dev)
            for i in range(length):
                acc ^= s_[:, i].to(torch.int64) << ((i & 15) << 1)
                acc ^= q[:, i].to(torch.int64) << ((i & 3) << 3)
            acc ^= ord("r")
            for k in range(12):
                e = ndig - 1 - k
                valid = e >= 0
                div = torch.pow(torch.tensor(10, dtype=torch.int64, device=dev), e.clamp(min=0))
                dig = (ids // div) % 10 + ord("0")
                acc ^= torch.where(valid, dig << (((k + 1) & 3) << 3), torch.zeros_like(dig))
            acc ^= torch.full_like(ids, ord("/")) << (((ndig + 1) & 3) << 3)
            acc ^= torch.full_like(ids, ord("0") + mate) << (((ndig + 2) & 3) << 3)
            acc &= 0xFFFFFFFF
            o["seed"][lo:lo + m] = torch.where(acc >= 2 ** 31, acc - 2 ** 32, acc).to(torch.int32)
    return out[0], out[1]
