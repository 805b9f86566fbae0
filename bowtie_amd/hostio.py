"""Python binding of the host I/O entry points of the C ABI (include/bowtie_amd.h, "host I/O"
section): read files -> ReadBatch, hit arrays -> the reference's SAM / default-format text.
The work is done by bowtie_amd/csrc/bt_io.cpp (C++); nothing here needs a GPU."""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi as A
from .aligner import BowtieAmdError, lib
from .reads import ReadBatch

FORMATS = {"fastq": A.BT_FMT_FASTQ, "fasta": A.BT_FMT_FASTA, "raw": A.BT_FMT_RAW, "cmdline": A.BT_FMT_CMDLINE,
           "fasta-cont": A.BT_FMT_FASTA_CONT, "tabbed": A.BT_FMT_TABBED}
QUALS = {"phred33": A.BT_QUAL_PHRED33, "phred64": A.BT_QUAL_PHRED64, "solexa": A.BT_QUAL_SOLEXA64,
         "int": 3, "int-solexa": 4}


class ReadInputError(ValueError):
    """Malformed read input; the text is the reference's own message."""


def read_batches(spec: str, fmt: str = "fastq", trim5: int = 0, trim3: int = 0, quals: str = "phred33",
                 seed: int = 0, skip: int = 0, upto: int = 0, max_reads: int = 1 << 20,
                 threads: int = 1, careful: bool = False, keep_raw: bool = False,
                 cont: Tuple[int, int] = (0, 0), mate: int = 0, interleaved: bool = False) -> Iterator[ReadBatch]:
    """Yield ReadBatch objects (copies) of up to max_reads reads each; keep_raw adds `.raw`, the list of
    the reads' records as they stood in the input.  fmt "tabbed" (--12): `mate` 2 delivers the records' second
    ends, and every batch carries `.n_paired`, the number of its reads whose record had one.  `interleaved`
    (--interleaved, FASTQ): `mate` 1 / 2 deliver the even / odd records."""
    L = lib()
    o = A.ReadOpts(FORMATS[fmt], trim5, trim3, QUALS[quals], seed, int(careful) | (2 if keep_raw else 0) | (4 if mate == 1 else 8 if mate == 2 else 0) | (16 if interleaved else 0), skip, upto,
                   cont[0], cont[1])
    h = C.c_void_p()
    rc = L.bt_reads_open(spec.encode(), C.byref(o), C.byref(h))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_reads_open")
    try:
        while True:
            rb = A.ReadBatchC()
            names, noff = C.c_void_p(), C.c_void_p()
            rc = L.bt_reads_next(h, max_reads, threads, C.byref(rb), C.byref(names), C.byref(noff))
            if rc == A.BT_ERR_READS:
                raise ReadInputError(L.bt_reads_error(h).decode(errors="replace"))
            if rc != A.BT_OK:
                raise BowtieAmdError(rc, "bt_reads_next")
            n = rb.n_reads
            if n == 0:
                return
            stride = rb.stride
            seq = np.ctypeslib.as_array(C.cast(rb.seq, C.POINTER(C.c_uint8)), shape=(n, stride)).copy()
            qual = np.ctypeslib.as_array(C.cast(rb.qual, C.POINTER(C.c_uint8)), shape=(n, stride)).copy()
            ln = np.ctypeslib.as_array(C.cast(rb.len, C.POINTER(C.c_uint16)), shape=(n,)).copy()
            sd = np.ctypeslib.as_array(C.cast(rb.seed, C.POINTER(C.c_uint32)), shape=(n,)).copy()
            off = np.ctypeslib.as_array(C.cast(noff, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
            blob = C.string_at(names, int(off[n]))
            nm = [blob[int(off[i]):int(off[i + 1])] for i in range(n)]
            rbatch = ReadBatch(seq, qual, ln, sd, nm)
            rbatch.n_paired = int(L.bt_reads_paired_count(h))
            if keep_raw:
                rp, ro = C.c_void_p(), C.c_void_p()
                if L.bt_reads_raw(h, C.byref(rp), C.byref(ro)) != A.BT_OK:
                    raise BowtieAmdError(A.BT_ERR_ARG, "bt_reads_raw")
                roff = np.ctypeslib.as_array(C.cast(ro, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
                rblob = C.string_at(rp, int(roff[n]))
                rbatch.raw = [rblob[int(roff[i]):int(roff[i + 1])] for i in range(n)]
            yield rbatch
    finally:
        L.bt_reads_close(h)


def read_all(spec: str, **kw) -> Optional[ReadBatch]:
    """All reads of `spec` as one batch (None if there are none)."""
    kw.setdefault("max_reads", 0x7FFFFFFF)
    bs = list(read_batches(spec, **kw))
    if not bs:
        return None
    assert len(bs) == 1
    return bs[0]


def out_opts(sam=False, full_ref=False, ref_idx=False, off_base=0, print_cost=False, show_seed=False, mapq=255,
             no_qname_trunc=False, no_unal=False, sam_nosq=False, khits=1, mhits=0xFFFFFFFF, all_hits=False,
             suppress: Sequence[int] = (), sample_max=False) -> A.OutOpts:
    mask = 0
    for f in suppress:
        mask |= 1 << (f - 1)
    return A.OutOpts(int(sam), int(full_ref), int(ref_idx), off_base, int(print_cost), int(show_seed), mapq,
                     int(no_qname_trunc), int(no_unal), int(sam_nosq), khits, mhits, int(all_hits), int(sample_max), mask)


def pack_hits(per_read, hit_cap: int):
    """[(hits: List[output.Hit], total, status)] -> the bt_hit_batch arrays."""
    n = len(per_read)
    hits = np.zeros(n * hit_cap, dtype=A.HIT_DTYPE)
    n_hits = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint8)
    pool: List[int] = []
    for i, (hs, tot, st) in enumerate(per_read):
        n_hits[i] = tot
        status[i] = st
        for k, h in enumerate(hs[:hit_cap]):
            r = hits[i * hit_cap + k]
            r["tidx"], r["toff"], r["oms"], r["cost"], r["stratum"], r["fw"] = h.tidx, h.toff, h.oms, h.cost, h.stratum, int(h.fw)
            r["mm_off"], r["nmm"] = len(pool), len(h.mms)
            r["pad"][0] = getattr(h, "mate", 0)
            pool.extend((p & 0x3FF) | (c << 12) for p, c in h.mms)
    return hits, n_hits, status, np.array(pool + [0], dtype=np.uint16)


def _names_blob(names: Sequence[bytes]):
    off = np.zeros(len(names) + 1, dtype=np.uint64)
    for i, x in enumerate(names):
        off[i + 1] = off[i] + len(x)
    return b"".join(names), off


def _refs(refnames: Sequence[str], reflens: Sequence[int]):
    arr = (C.c_char_p * max(1, len(refnames)))(*[r.encode() for r in refnames])
    lens = np.asarray(list(reflens) + [0], dtype=np.uint32)
    return arr, lens


def format_hits(batch: ReadBatch, hits, n_hits, status, mm_pool, hit_cap: int, refnames: Sequence[str],
                reflens: Sequence[int], opts: A.OutOpts) -> Tuple[bytes, A.OutTally]:
    L = lib()
    seq = np.ascontiguousarray(batch.seq, dtype=np.uint8)
    qual = np.ascontiguousarray(batch.qual, dtype=np.uint8)
    ln = np.ascontiguousarray(batch.len, dtype=np.uint16)
    sd = np.ascontiguousarray(batch.seed, dtype=np.uint32)
    rb = A.ReadBatchC(batch.n, batch.stride, seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, sd.ctypes.data)
    hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, mm_pool.ctypes.data,
                     len(mm_pool), 0)
    blob, off = _names_blob(batch.names)
    arr, lens = _refs(refnames, reflens)
    text, tlen, tally = C.c_void_p(), C.c_size_t(), A.OutTally()
    rc = L.bt_format_hits(C.byref(rb), blob, off.ctypes.data, C.byref(hb), arr, lens.ctypes.data, len(refnames),
                          C.byref(opts), C.byref(text), C.byref(tlen), C.byref(tally))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_format_hits")
    try:
        return C.string_at(text, tlen.value), tally
    finally:
        L.bt_text_free(text)


def format_pairs(b1: ReadBatch, b2: ReadBatch, hits, n_hits, status, mm_pool, hit_cap: int, refnames: Sequence[str],
                 reflens: Sequence[int], opts: A.OutOpts) -> Tuple[bytes, A.OutTally]:
    """bt_format_pairs: the paired hit layout of bt_align_pairs -> the reference's text."""
    from .aligner import pack_batch
    L = lib()
    k1, rb1 = pack_batch(b1)
    k2, rb2 = pack_batch(b2)
    hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, mm_pool.ctypes.data, len(mm_pool), 0)
    blob1, off1 = _names_blob(b1.names)
    blob2, off2 = _names_blob(b2.names)
    arr, lens = _refs(refnames, reflens)
    text, tlen, tally = C.c_void_p(), C.c_size_t(), A.OutTally()
    L.bt_format_pairs.argtypes = None
    rc = L.bt_format_pairs(C.byref(rb1), blob1, C.c_void_p(off1.ctypes.data), C.byref(rb2), blob2, C.c_void_p(off2.ctypes.data),
                           C.byref(hb), arr, C.c_void_p(lens.ctypes.data), C.c_uint32(len(refnames)), C.byref(opts),
                           C.byref(text), C.byref(tlen), C.byref(tally))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_format_pairs")
    try:
        return C.string_at(text, tlen.value), tally
    finally:
        L.bt_text_free(text)


def sam_header(refnames: Sequence[str], reflens: Sequence[int], opts: A.OutOpts, cmdline: str,
               rgline: Optional[str] = None) -> bytes:
    L = lib()
    arr, lens = _refs(refnames, reflens)
    text, tlen = C.c_void_p(), C.c_size_t()
    rc = L.bt_format_sam_header(arr, lens.ctypes.data, len(refnames), C.byref(opts), cmdline.encode(),
                                rgline.encode() if rgline else None, C.byref(text), C.byref(tlen))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_format_sam_header")
    try:
        return C.string_at(text, tlen.value)
    finally:
        L.bt_text_free(text)


def summary(tally: A.OutTally) -> str:
    L = lib()
    text, tlen = C.c_void_p(), C.c_size_t()
    rc = L.bt_format_summary(C.byref(tally), C.byref(text), C.byref(tlen))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_format_summary")
    try:
        return C.string_at(text, tlen.value).decode()
    finally:
        L.bt_text_free(text)
