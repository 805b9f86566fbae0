"""Host-side read input for the hot path: what the reference's PatternSourcePerThread hands a
worker (read.h:42-273) -- patFw codes, Phred+33 quals, name and the per-read RNG seed.

Mirrors (behaviour, not code):
  * FastqPatternSource::parse      pat.cpp:862-975   (name = whole header line, '.' -> N,
                                                       non-alphabetic sequence chars dropped)
  * asc2dna                        alphabet.cpp:107   (ACGT any case -> 0-3, other letters -> 4)
  * genRandSeed                    pat.cpp:21-57
This is the round-1 host surface (plain Python/numpy); the C-ABI in include/bowtie_amd.h takes
the packed SoA batch this module produces.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass
from typing import Iterable, List, Sequence

import numpy as np

_ASC2DNA = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _ASC2DNA[ord(_c)] = _i
    _ASC2DNA[ord(_c.lower())] = _i
_DNA2ASC = np.frombuffer(b"ACGTN", dtype=np.uint8)


@dataclass
class Read:
    name: bytes
    seq: np.ndarray      # uint8 codes 0..4
    qual: bytes          # Phred+33 ASCII

    def __len__(self) -> int:
        return len(self.seq)


def encode_seq(s: bytes) -> np.ndarray:
    a = np.frombuffer(s.replace(b".", b"N"), dtype=np.uint8)
    alpha = ((a >= 65) & (a <= 90)) | ((a >= 97) & (a <= 122))
    return _ASC2DNA[a[alpha]]


def decode_seq(codes: np.ndarray) -> bytes:
    return _DNA2ASC[np.asarray(codes, dtype=np.uint8)].tobytes()


def _open(path):
    if str(path).endswith(".gz"):
        return gzip.open(path, "rb")
    return open(path, "rb")


def parse_fastq(path, trim5: int = 0, trim3: int = 0) -> List[Read]:
    """FASTQ reads, 4 lines per record (pat.cpp:797-975)."""
    reads: List[Read] = []
    with _open(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    i = 0
    while i < len(lines) and lines[i].strip(b"\r") == b"":
        i += 1
    rdid = 0
    while i < len(lines):
        if i + 3 >= len(lines):
            break                       # truncated final record: dropped, as the reference does
        hdr = lines[i].rstrip(b"\r")
        if not hdr.startswith(b"@"):
            raise ValueError("reads file does not look like a FASTQ file")
        seq = encode_seq(lines[i + 1].rstrip(b"\r"))
        qual = lines[i + 3].rstrip(b"\r")
        if trim5:
            seq = seq[trim5:]
            qual = qual[trim5:]
        if trim3:
            seq = seq[: max(0, len(seq) - trim3)]
            qual = qual[: max(0, len(qual) - trim3)]
        if len(qual) < len(seq):
            raise ValueError("fewer quality values than bases for read %r" % hdr)
        if len(qual) > len(seq):
            raise ValueError("more quality values than bases for read %r" % hdr)
        name = hdr[1:]
        if not name:
            name = str(rdid).encode()
        reads.append(Read(name, seq, qual))
        rdid += 1
        i += 4
    return reads


def parse_fasta(path, default_qual: bytes = b"I") -> List[Read]:
    """FASTA reads (one record per '>' header); quals default to 'I' (pat.cpp FastaPatternSource)."""
    reads: List[Read] = []
    name = None
    chunks: List[bytes] = []
    rdid = 0

    def flush():
        nonlocal rdid, name, chunks
        if name is None:
            return
        seq = encode_seq(b"".join(chunks))
        nm = name if name else str(rdid).encode()
        reads.append(Read(nm, seq, default_qual * len(seq)))
        rdid += 1

    with _open(path) as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                flush()
                name = line[1:]
                chunks = []
            else:
                chunks.append(line)
    flush()
    return reads


@dataclass
class ReadBatch:
    """SoA batch = bt_read_batch (include/bowtie_amd.h)."""
    seq: np.ndarray     # [n, stride] uint8
    qual: np.ndarray    # [n, stride] uint8
    len: np.ndarray     # [n] uint16
    seed: np.ndarray    # [n] uint32
    names: List[bytes]

    @property
    def n(self) -> int:
        return int(self.seq.shape[0])

    @property
    def stride(self) -> int:
        return int(self.seq.shape[1])


def rand_seeds(seq: np.ndarray, qual: np.ndarray, lens: np.ndarray, names: Sequence[bytes],
               global_seed: int = 0) -> np.ndarray:
    """genRandSeed (pat.cpp:21-57), vectorised over a padded batch."""
    n, stride = seq.shape
    pos = np.arange(stride, dtype=np.uint32)
    valid = pos[None, :] < lens[:, None].astype(np.uint32)
    base = np.uint32(((global_seed + 101) * 59 * 61 * 67 * 71 * 73 * 79 * 83) & 0xFFFFFFFF)
    s = (seq.astype(np.uint32) << ((pos & 15) << 1)[None, :])
    s = np.where(valid, s, 0)
    q = (qual.astype(np.uint32) << ((pos & 3) << 3)[None, :])
    q = np.where(valid, q, 0)
    out = np.bitwise_xor.reduce(s, axis=1) ^ np.bitwise_xor.reduce(q, axis=1) ^ base
    # names
    maxn = max((len(x) for x in names), default=0)
    if maxn:
        nm = np.zeros((n, maxn), dtype=np.uint32)
        for i, x in enumerate(names):
            nm[i, : len(x)] = np.frombuffer(x, dtype=np.uint8)
        npos = np.arange(maxn, dtype=np.uint32)
        out ^= np.bitwise_xor.reduce(nm << ((npos & 3) << 3)[None, :], axis=1)
    return out.astype(np.uint32)


def pack_reads(reads: Iterable[Read], global_seed: int = 0, stride: int | None = None) -> ReadBatch:
    reads = list(reads)
    n = len(reads)
    maxlen = max((len(r) for r in reads), default=1)
    if stride is None:
        stride = max(16, (maxlen + 15) & ~15)     # rows 16-byte aligned (the kernel fetches 16-byte windows)
    seq = np.full((n, stride), 4, dtype=np.uint8)
    qual = np.full((n, stride), 33, dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    names = []
    for i, r in enumerate(reads):
        L = len(r)
        seq[i, :L] = r.seq
        qual[i, :L] = np.frombuffer(r.qual, dtype=np.uint8)
        lens[i] = L
        names.append(r.name)
    seed = rand_seeds(seq, qual, lens, names, global_seed)
    return ReadBatch(seq, qual, lens, seed, names)
