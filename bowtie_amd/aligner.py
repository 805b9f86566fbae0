"""Host-side binding of the C ABI (include/bowtie_amd.h) -- the drop-in boundary.

`Index` mirrors the reference's `Ebwt` objects (ebwt_search.cpp:3120-3155) and `Aligner` the
per-thread worker loop (ebwt_search.cpp:1130/1606/2056/2378): reads in, `Hit` records out.
There is no fallback: if bowtie_amd/libbowtie_amd.so (HIP, gfx950) is missing or no GPU is
present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _abi as A
from .output import Hit
from .reads import ReadBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("BT_LIB", "libbowtie_amd.so"))
EXPORTS = ["bt_policy_default", "bt_has_pe_v1", "bt_index_load", "bt_index_info_get", "bt_index_refname",
           "bt_index_reflen", "bt_index_free", "bt_index_restore_text", "bt_index_digest", "bt_index_needs_rows64", "bt_ctx_create", "bt_ctx_destroy", "bt_align_batch",
           "bt_align_batch_device", "bt_index_load_reference", "bt_align_pairs", "bt_align_pairs_device", "bt_align_stream_submit", "bt_align_stream_room", "bt_align_stream_collect", "bt_align_stream_tick", "bt_host_alloc", "bt_host_free", "bt_ctx_sync", "bt_ctx_set_carry", "bt_ctx_set_max_read_len", "bt_ctx_span_ms", "bt_ctx_launch_ms", "bt_ctx_last_carried", "bt_ctx_last_kernel_ms", "bt_ctx_last_kernel_name", "bt_ctx_last_mm_used", "bt_ctx_jump_counts", "bt_index_jump_bytes", "bt_ctx_last_retried", "bt_ctx_set_locus", "bt_ctx_get_locus", "bt_index_locus_bytes", "bt_index_locus_build_seconds", "bt_index_locus_copy",
           "bt_ctx_counts", "bt_ctx_set_iters_buffer", "bt_ctx_prof_sections", "bt_strerror", "bt_version", "bt_rows64", "bt_index_len64", "bt_probe_rank", "bt_probe_rank64", "bt_probe_chase", "bt_bench_gather",
           "bt_reads_open", "bt_reads_next", "bt_reads_raw", "bt_reads_paired_count", "bt_reads_error", "bt_reads_close", "bt_format_hits", "bt_format_pairs",
           "bt_format_sam_header", "bt_format_summary", "bt_text_free"]
_lib = None


class BowtieAmdError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        super().__init__("%s: %s (code %d)" % (what, strerror(code), code))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("bowtie_amd: %s is missing -- build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.bt_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bt_host_alloc.argtypes = [C.c_size_t]
        L.bt_host_alloc.restype = C.c_void_p
        L.bt_host_free.argtypes = [C.c_void_p]
        L.bt_host_free.restype = None
        L.bt_align_stream_room.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.bt_index_info_get.argtypes = [C.c_void_p, C.POINTER(A.IndexInfo)]
        L.bt_index_info_get.restype = None
        L.bt_index_refname.argtypes = [C.c_void_p, C.c_uint32]
        L.bt_index_refname.restype = C.c_char_p
        L.bt_index_reflen.argtypes = [C.c_void_p, C.c_uint32]
        L.bt_index_reflen.restype = C.c_uint32
        L.bt_index_restore_text.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        L.bt_index_digest.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
        L.bt_index_free.argtypes = [C.c_void_p]
        L.bt_index_free.restype = None
        L.bt_ctx_create.argtypes = [C.c_void_p, C.POINTER(A.Policy), C.c_void_p, C.POINTER(C.c_void_p)]
        L.bt_ctx_destroy.argtypes = [C.c_void_p]
        L.bt_ctx_destroy.restype = None
        L.bt_align_batch.argtypes = [C.c_void_p, C.POINTER(A.ReadBatchC), C.POINTER(A.HitBatchC),
                                     C.POINTER(A.OpCounts)]
        L.bt_align_batch_device.argtypes = [C.c_void_p, C.POINTER(A.ReadBatchC), C.POINTER(A.HitBatchC),
                                            C.c_void_p]
        L.bt_index_load_reference.argtypes = [C.c_void_p]
        L.bt_align_pairs.argtypes = [C.c_void_p, C.POINTER(A.ReadBatchC), C.POINTER(A.ReadBatchC), C.POINTER(A.HitBatchC),
                                     C.POINTER(A.OpCounts)]
        L.bt_align_pairs_device.argtypes = [C.c_void_p, C.POINTER(A.ReadBatchC), C.POINTER(A.ReadBatchC),
                                            C.POINTER(A.HitBatchC), C.c_void_p]
        L.bt_ctx_sync.argtypes = [C.c_void_p]
        L.bt_ctx_last_kernel_ms.argtypes = [C.c_void_p]
        L.bt_ctx_set_carry.argtypes = [C.c_void_p, C.c_int]
        L.bt_ctx_set_max_read_len.argtypes = [C.c_void_p, C.c_uint32]
        L.bt_align_stream_submit.argtypes = [C.c_void_p, C.POINTER(A.ReadBatchC), C.POINTER(A.HitBatchC), C.c_void_p]
        L.bt_align_stream_collect.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]
        L.bt_align_stream_tick.argtypes = [C.c_void_p, C.c_uint32]
        L.bt_ctx_set_locus.argtypes = [C.c_void_p, C.c_int]
        L.bt_ctx_get_locus.argtypes = [C.c_void_p]
        L.bt_index_locus_bytes.argtypes = [C.c_void_p]
        L.bt_index_locus_bytes.restype = C.c_uint64
        if hasattr(L, "bt_index_jump_bytes"):          # (a library of an earlier round loaded with BT_LIB has none)
            L.bt_index_jump_bytes.restype = C.c_uint64
            L.bt_index_jump_bytes.argtypes = [C.c_void_p]
            L.bt_ctx_jump_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.bt_ctx_jump_counts.restype = None
        L.bt_index_locus_build_seconds.argtypes = [C.c_void_p]
        L.bt_index_locus_build_seconds.restype = C.c_double
        L.bt_index_locus_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bt_ctx_span_ms.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.bt_ctx_span_ms.restype = C.c_float
        L.bt_ctx_launch_ms.argtypes = [C.c_void_p, C.c_int]
        L.bt_ctx_launch_ms.restype = C.c_float
        L.bt_ctx_last_carried.argtypes = [C.c_void_p]
        L.bt_ctx_last_carried.restype = C.c_uint32
        L.bt_ctx_last_kernel_name.argtypes = [C.c_void_p]
        L.bt_ctx_last_kernel_name.restype = C.c_char_p
        L.bt_ctx_last_kernel_ms.restype = C.c_float
        L.bt_ctx_last_mm_used.argtypes = [C.c_void_p]
        L.bt_ctx_last_mm_used.restype = C.c_uint32
        L.bt_ctx_last_retried.argtypes = [C.c_void_p]
        L.bt_ctx_last_retried.restype = C.c_uint32
        L.bt_ctx_set_iters_buffer.argtypes = [C.c_void_p, C.c_void_p]
        L.bt_ctx_set_iters_buffer.restype = None
        L.bt_ctx_counts.argtypes = [C.c_void_p, C.POINTER(A.OpCounts), C.c_int]
        L.bt_strerror.argtypes = [C.c_int]
        L.bt_strerror.restype = C.c_char_p
        L.bt_version.restype = C.c_char_p
        L.bt_probe_rank.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.bt_probe_rank64.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.bt_index_len64.argtypes = [C.c_void_p]
        L.bt_index_len64.restype = C.c_uint64
        L.bt_probe_chase.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.bt_policy_default.argtypes = [C.POINTER(A.Policy)]
        L.bt_policy_default.restype = None
        L.bt_reads_open.argtypes = [C.c_char_p, C.POINTER(A.ReadOpts), C.POINTER(C.c_void_p)]
        L.bt_reads_next.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(A.ReadBatchC),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bt_reads_raw.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bt_reads_paired_count.argtypes = [C.c_void_p]
        L.bt_reads_paired_count.restype = C.c_uint32
        L.bt_reads_error.argtypes = [C.c_void_p]
        L.bt_reads_error.restype = C.c_char_p
        L.bt_reads_close.argtypes = [C.c_void_p]
        L.bt_reads_close.restype = None
        L.bt_format_hits.argtypes = [C.POINTER(A.ReadBatchC), C.c_void_p, C.c_void_p, C.POINTER(A.HitBatchC),
                                     C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.POINTER(A.OutOpts),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(A.OutTally)]
        L.bt_format_sam_header.argtypes = [C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.POINTER(A.OutOpts),
                                           C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.bt_format_summary.argtypes = [C.POINTER(A.OutTally), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.bt_text_free.argtypes = [C.c_void_p]
        L.bt_text_free.restype = None
        _lib = L
    return _lib


def strerror(code: int) -> str:
    return lib().bt_strerror(code).decode()


def index_digest(base: str, mirror: bool = False):
    """bt_index_digest: [variant | swapped, len, digests of ebwt, ftab, eftab, offs, plen+rstarts, scalars]."""
    out = (C.c_uint64 * 8)()
    rc = lib().bt_index_digest(base.encode(), int(mirror), out)
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_index_digest(%s)" % base)
    return [int(x) for x in out]


def restore_text(base: str) -> np.ndarray:
    """Joined reference text (codes 0..3) of an index, recovered on the host (bowtie-inspect's job)."""
    out = np.zeros(index_digest(base)[1], dtype=np.uint8)
    rc = lib().bt_index_restore_text(base.encode(), out.ctypes.data, len(out))
    if rc != A.BT_OK:
        raise BowtieAmdError(rc, "bt_index_restore_text(%s)" % base)
    return out


class Index:
    """Device-resident fw (+ mirror) index image: Ebwt ctor + loadIntoMemory + H2D upload."""

    def __init__(self, base: str, need_mirror: bool = True, offrate: int = -1, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().bt_index_load(base.encode(), int(need_mirror), offrate, device, C.byref(self._h))
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_index_load(%s)" % base)
        info = A.IndexInfo()
        lib().bt_index_info_get(self._h, C.byref(info))
        self.info = info
        self.refnames = [lib().bt_index_refname(self._h, i).decode() for i in range(info.n_pat)]
        self.reflens = [int(lib().bt_index_reflen(self._h, i)) for i in range(info.n_pat)]

    def close(self):
        if self._h:
            lib().bt_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_hits(n: int, hit_cap: int, hits: np.ndarray, n_hits: np.ndarray, status: np.ndarray,
                mm_pool: np.ndarray, khits: int, mhits: int, all_hits: bool, sample_max: bool = False):
    """-> per read (hits: List[Hit], hitsForThisRead, status), applying finishRead's rules
    (hit.h:741-786): a read over the -m ceiling reports nothing (with -M: keeps the first mhits
    hits, one of which the output stage samples); otherwise the first -k hits."""
    out = []
    lim = hit_cap if all_hits else min(hit_cap, khits)
    for i in range(n):
        tot = int(n_hits[i])
        hs: List[Hit] = []
        if tot <= mhits or sample_max:
            for k in range(min(mhits, hit_cap) if tot > mhits else min(tot, lim)):
                h = hits[i * hit_cap + k]
                off, nmm = int(h["mm_off"]), int(h["nmm"])
                mms = [(int(e) & 0x3FF, (int(e) >> 12) & 3) for e in mm_pool[off:off + nmm]]
                hs.append(Hit(int(h["tidx"]), int(h["toff"]), int(h["oms"]), int(h["cost"]),
                              int(h["stratum"]), bool(h["fw"]), mms))
        out.append((hs, tot, int(status[i])))
    return out


def pack_batch(batch: ReadBatch):
    """ReadBatch -> (arrays kept alive, ReadBatchC over them)"""
    seq = np.ascontiguousarray(batch.seq, dtype=np.uint8)
    qual = np.ascontiguousarray(batch.qual, dtype=np.uint8)
    ln = np.ascontiguousarray(batch.len, dtype=np.uint16)
    seed = np.ascontiguousarray(batch.seed, dtype=np.uint32)
    return (seq, qual, ln, seed), A.ReadBatchC(batch.n, batch.stride, seq.ctypes.data, qual.ctypes.data,
                                               ln.ctypes.data, seed.ctypes.data)


def pair_hit_cap(pol) -> int:
    """Hit slots per pair: two per alignment wanted; with -M the first mhits pairs are kept for the sampling."""
    if pol.all_hits:
        return 128
    want = 2 * int(pol.khits)
    if pol.sample_max:
        want = max(want, 2 * int(pol.mhits))
    return max(2, min(want, 128))


def unpack_pair_hits(n: int, hit_cap: int, hits, n_hits, status, mm_pool, pol):
    """Paired-end results: per pair (hits: upstream mate, downstream mate, ..., hitsForThisRead, status),
    finishRead's rules with the doubled -k / -m of createMult(2) (hit.h:741-786)."""
    out = []
    mhits, khits = int(pol.mhits), int(pol.khits)
    maxv = 0xFFFFFFFF if mhits == 0xFFFFFFFF else 2 * mhits
    lim = hit_cap if pol.all_hits else min(hit_cap, 2 * khits)
    for i in range(n):
        tot = int(n_hits[i])
        hs: List[Hit] = []
        if tot <= maxv or pol.sample_max:
            # over the -m ceiling with -M: the first mhits pairs were buffered, one of them gets printed
            for k in range(min(tot, lim) if tot <= maxv else min(maxv, hit_cap) & ~1):
                h = hits[i * hit_cap + k]
                off, nmm = int(h["mm_off"]), int(h["nmm"])
                mms = [(int(e) & 0x3FF, (int(e) >> 12) & 3) for e in mm_pool[off:off + nmm]]
                hs.append(Hit(int(h["tidx"]), int(h["toff"]), int(h["oms"]), int(h["cost"]), int(h["stratum"]),
                              bool(h["fw"]), mms, int(h["pad"][0])))
        out.append((hs, tot, int(status[i])))
    return out


class Aligner:
    """One per GPU; `align(batch)` = the worker loop body over a batch of reads."""

    def __init__(self, index: Index, policy: A.Policy, stream: Optional[int] = None):
        self.index = index
        self.policy = policy
        self._h = C.c_void_p()
        rc = lib().bt_ctx_create(index._h, C.byref(policy), C.c_void_p(stream) if stream else None,
                                 C.byref(self._h))
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_ctx_create")

    def close(self):
        if self._h:
            lib().bt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_hit_cap(self) -> int:
        if self.policy.sample_max:      # -M: the first mhits hits are kept for sampling
            return max(1, min(max(int(self.policy.khits), int(self.policy.mhits)), 64))
        return 64 if self.policy.all_hits else max(1, min(int(self.policy.khits), 64))

    def align(self, batch: ReadBatch, hit_cap: Optional[int] = None, mm_per_hit: int = 8,
              counts: Optional[A.OpCounts] = None):
        """Host arrays in, host arrays out (bt_align_batch).  Returns unpack_hits()'s list."""
        n = batch.n
        hit_cap = hit_cap or self.default_hit_cap()
        seq = np.ascontiguousarray(batch.seq, dtype=np.uint8)
        qual = np.ascontiguousarray(batch.qual, dtype=np.uint8)
        ln = np.ascontiguousarray(batch.len, dtype=np.uint16)
        seed = np.ascontiguousarray(batch.seed, dtype=np.uint32)
        hits = np.zeros(n * hit_cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint8)
        pool = np.zeros(max(1, n * hit_cap * mm_per_hit), dtype=np.uint16)
        rb = A.ReadBatchC(n, batch.stride, seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, seed.ctypes.data)
        hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data,
                         pool.ctypes.data, len(pool), 0)
        rc = lib().bt_align_batch(self._h, C.byref(rb), C.byref(hb),
                                  C.byref(counts) if counts is not None else None)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_align_batch")
        self.last_kernel_ms = float(lib().bt_ctx_last_kernel_ms(self._h))
        self.last_retried = int(lib().bt_ctx_last_retried(self._h))
        return unpack_hits(n, hit_cap, hits, n_hits, status, pool, int(self.policy.khits),
                           int(self.policy.mhits), bool(self.policy.all_hits),
                           sample_max=bool(self.policy.sample_max))

    def align_pairs(self, b1: ReadBatch, b2: ReadBatch, hit_cap: Optional[int] = None, mm_per_hit: int = 8,
                    counts: Optional[A.OpCounts] = None):
        """Paired-end (bt_align_pairs): per pair (hits: upstream mate, downstream mate, ...,
        hitsForThisRead, status).  Loads the 2-bit reference into HBM on first use."""
        rc = lib().bt_index_load_reference(self.index._h)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_index_load_reference")
        n = b1.n
        hit_cap = hit_cap or pair_hit_cap(self.policy)
        k1, rb1 = pack_batch(b1)
        k2, rb2 = pack_batch(b2)
        hits = np.zeros(n * hit_cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint8)
        pool = np.zeros(max(1, n * hit_cap * mm_per_hit), dtype=np.uint16)
        hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
        rc = lib().bt_align_pairs(self._h, C.byref(rb1), C.byref(rb2), C.byref(hb),
                                  C.byref(counts) if counts is not None else None)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_align_pairs")
        self.last_kernel_ms = float(lib().bt_ctx_last_kernel_ms(self._h))
        self.last_retried = int(lib().bt_ctx_last_retried(self._h))
        return unpack_pair_hits(n, hit_cap, hits, n_hits, status, pool, self.policy)

    def probe_rank(self, rows: np.ndarray, mirror: bool = False, sides: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """LF(row, ACGT) and the BWT character at `row`: from the 32-byte rank blocks the search kernels query, or
        (sides=True) straight from the index files' 224-symbol side layout."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        lf = np.zeros((len(rows), 4), dtype=np.uint32)
        L = np.zeros(len(rows), dtype=np.uint8)
        rc = lib().bt_probe_rank(self._h, int(mirror) | (2 if sides else 0), rows.ctypes.data, len(rows), lf.ctypes.data, L.ctypes.data)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_probe_rank")
        return lf, L

    def probe_rank64(self, rows: np.ndarray, mirror: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """the same from the rank blocks with rows as 64-bit numbers (either library; the only rank probe of libbowtie_amd_l.so)"""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        lf = np.zeros((len(rows), 4), dtype=np.uint64)
        L = np.zeros(len(rows), dtype=np.uint8)
        rc = lib().bt_probe_rank64(self._h, int(mirror), rows.ctypes.data, len(rows), lf.ctypes.data, L.ctypes.data)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_probe_rank64")
        return lf, L

    def probe_chase(self, rows: np.ndarray, qlen: int, mirror: bool = False):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        j = np.zeros(len(rows), dtype=np.uint32)
        t = np.zeros(len(rows), dtype=np.uint32)
        o = np.zeros(len(rows), dtype=np.uint32)
        rc = lib().bt_probe_chase(self._h, int(mirror), rows.ctypes.data, len(rows), qlen,
                                  j.ctypes.data, t.ctypes.data, o.ctypes.data)
        if rc != A.BT_OK:
            raise BowtieAmdError(rc, "bt_probe_chase")
        return j, t, o
