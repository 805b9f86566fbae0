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
    stride = stride or max(4, (length + 3) & ~3)
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
    with open(path, "wb") as f:
        for i in range(batch.n):
            L = int(batch.len[i])
            f.write(b"@" + batch.names[i] + b"\n" + asc[batch.seq[i, :L]].tobytes() + b"\n+\n" +
                    batch.qual[i, :L].tobytes() + b"\n")
