"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (the checker).

Never imported by the bowtie_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
BTO_MAXMM = 64


from bowtie_amd._abi import Policy, make_policy   # noqa: E402  (bt_policy, include/bowtie_amd.h)


class OpCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames", "lane_iters", "same_pair", "rescans", "cand_scans", "wave_rounds", "fetches",
                 "loc_lfex", "loc_lf1", "loc_chase", "loc_records", "loc_windows")]


class OHit(C.Structure):
    _fields_ = [("tidx", C.c_uint32), ("toff", C.c_uint32), ("oms", C.c_uint32),
                ("cost", C.c_uint16), ("stratum", C.c_uint8), ("fw", C.c_uint8),
                ("nmm", C.c_uint16), ("mate", C.c_uint16), ("mm", C.c_uint16 * BTO_MAXMM)]


class OIndex(C.Structure):
    _fields_ = [("len", C.c_uint32), ("bwtLen", C.c_uint32), ("sideSz", C.c_uint32),
                ("sideBwtSz", C.c_uint32), ("sideBwtLen", C.c_uint32), ("numSides", C.c_uint32),
                ("ebwtTotLen", C.c_uint32), ("ftabChars", C.c_uint32), ("ftabLen", C.c_uint32),
                ("eftabLen", C.c_uint32), ("offRate", C.c_uint32), ("offMask", C.c_uint32),
                ("offsLen", C.c_uint32), ("nPat", C.c_uint32), ("nFrag", C.c_uint32),
                ("zOff", C.c_uint32), ("zEbwtByteOff", C.c_uint32), ("zEbwtBpOff", C.c_int32),
                ("fw", C.c_int32), ("fchr", C.c_uint32 * 5),
                ("ebwt", C.POINTER(C.c_uint8)), ("plen", C.POINTER(C.c_uint32)),
                ("rstarts", C.POINTER(C.c_uint32)), ("ftab", C.POINTER(C.c_uint32)),
                ("eftab", C.POINTER(C.c_uint32)), ("offs", C.POINTER(C.c_uint32)),
                ("refnames", C.POINTER(C.c_char_p)), ("wide", C.c_int32)]




_lib = None


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "port"])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or \
                os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(ORACLE_DIR, f))
                                                 for f in ("bt_oracle.c", "bt_oracle_best.c", "bt_oracle.h")):
            build()
        L = C.CDLL(LIB_PATH)
        L.bto_index_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(OIndex)]
        L.bto_rank4.argtypes = [C.POINTER(OIndex), C.c_uint32, C.POINTER(C.c_uint32)]
        L.bto_rowL.argtypes = [C.POINTER(OIndex), C.c_uint32]
        L.bto_ftab_hi.argtypes = [C.POINTER(OIndex), C.c_uint32]
        L.bto_ftab_hi.restype = C.c_uint32
        L.bto_ftab_lo.argtypes = [C.POINTER(OIndex), C.c_uint32]
        L.bto_ftab_lo.restype = C.c_uint32
        L.bto_chase.argtypes = [C.POINTER(OIndex), C.c_uint32, C.POINTER(C.c_uint32)]
        L.bto_chase.restype = C.c_uint32
        L.bto_joined_to_text.argtypes = [C.POINTER(OIndex), C.c_uint32, C.c_uint32] + \
            [C.POINTER(C.c_uint32)] * 3
        L.bto_restore_text.argtypes = [C.POINTER(OIndex), C.c_void_p]
        L.bto_restore_text.restype = None
        L.bto_rand_seed.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint32]
        L.bto_rand_seed.restype = C.c_uint32
        L.bto_align_read.argtypes = [C.POINTER(OIndex), C.POINTER(OIndex), C.POINTER(Policy),
                                     C.c_char_p, C.c_char_p, C.c_int, C.c_uint32,
                                     C.POINTER(OHit), C.c_int, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(OpCounts)]
        L.bto_refs_build.argtypes = [C.POINTER(OIndex)]
        L.bto_refs_build.restype = C.c_void_p
        L.bto_refs_free.argtypes = [C.c_void_p]
        L.bto_align_pair_v1.argtypes = [C.POINTER(OIndex), C.POINTER(OIndex), C.c_void_p, C.POINTER(Policy),
                                        C.c_char_p, C.c_char_p, C.c_int, C.c_uint32,
                                        C.c_char_p, C.c_char_p, C.c_int, C.c_uint32,
                                        C.POINTER(OHit), C.c_int, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(OpCounts)]
        L.bto_align_pair_best.argtypes = [C.POINTER(OIndex), C.POINTER(OIndex), C.c_void_p, C.POINTER(Policy),
                                          C.c_char_p, C.c_char_p, C.c_int, C.c_uint32,
                                          C.c_char_p, C.c_char_p, C.c_int, C.c_uint32,
                                          C.POINTER(OHit), C.c_int, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32), C.POINTER(OpCounts)]
        _lib = L
    return _lib


class OracleIndex:
    """fw (+ mirror) index pair loaded by the oracle's own .ebwt parser."""

    def __init__(self, base: str, need_mirror: bool = True, wide: bool = False):
        """wide: restate bowtie-align-l (the 64-bit build) on this index's arrays (see bto_index.wide)."""
        L = lib()
        self.fw = OIndex()
        rc = L.bto_index_load(base.encode(), 1, C.byref(self.fw))
        if rc:
            raise IOError("oracle: cannot load %s (rc=%d)" % (base, rc))
        self.bw = None
        if need_mirror:
            self.bw = OIndex()
            rc = L.bto_index_load((base + ".rev").encode(), 0, C.byref(self.bw))
            if rc:
                raise IOError("oracle: cannot load %s.rev (rc=%d)" % (base, rc))
        self.fw.wide = int(wide)
        if self.bw is not None:
            self.bw.wide = int(wide)
        self.refnames = [self.fw.refnames[i].decode() for i in range(self.fw.nPat)]
        self.reflens = [int(self.fw.plen[i]) for i in range(self.fw.nPat)]

    def joined_text(self) -> np.ndarray:
        """The joined reference text (codes 0..3) recovered from the fw index."""
        out = np.zeros(self.fw.len, dtype=np.uint8)
        lib().bto_restore_text(C.byref(self.fw), out.ctypes.data)
        return out

    def ix(self, mirror: bool) -> OIndex:
        return self.bw if mirror else self.fw

    def rank4(self, row: int, mirror: bool = False):
        out = (C.c_uint32 * 4)()
        lib().bto_rank4(C.byref(self.ix(mirror)), row, out)
        return list(out), lib().bto_rowL(C.byref(self.ix(mirror)), row)

    def chase(self, row: int, mirror: bool = False) -> Tuple[int, int]:
        j = C.c_uint32()
        off = lib().bto_chase(C.byref(self.ix(mirror)), row, C.byref(j))
        return int(off), int(j.value)

    def joined_to_text(self, qlen: int, off: int, mirror: bool = False):
        t, o, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        ok = lib().bto_joined_to_text(C.byref(self.ix(mirror)), qlen, off, C.byref(t), C.byref(o), C.byref(l))
        return (int(t.value), int(o.value)) if ok else (0xFFFFFFFF, 0)

    def align(self, pol: Policy, seq: np.ndarray, qual: bytes, seed: int, cap: int = 16,
              counts: Optional[OpCounts] = None):
        """-> (hits[list of dict], n_hits_total, status)"""
        hits = (OHit * cap)()
        tot, st = C.c_uint32(), C.c_uint32()
        n = lib().bto_align_read(C.byref(self.fw), C.byref(self.bw) if self.bw is not None else None,
                                 C.byref(pol), seq.tobytes(), bytes(qual), len(seq), seed, hits, cap,
                                 C.byref(tot), C.byref(st), C.byref(counts) if counts is not None else None)
        if n < 0:
            raise ValueError("oracle error %d" % -n)
        return self._unpack(hits, n), int(tot.value), int(st.value)

    @staticmethod
    def _unpack(hits, n):
        out = []
        for i in range(n):
            h = hits[i]
            out.append(dict(tidx=h.tidx, toff=h.toff, oms=h.oms, cost=h.cost, stratum=h.stratum,
                            fw=bool(h.fw), mate=int(h.mate),
                            mms=[(h.mm[k] & 0x3FF, (h.mm[k] >> 12) & 3) for k in range(h.nmm)]))
        return out

    def align_pair(self, pol: Policy, seq1, qual1: bytes, seed1: int, seq2, qual2: bytes, seed2: int,
                   cap: int = 16, counts: Optional[OpCounts] = None, v1: bool = False):
        """PairedBWAlignerV2 (--best) or, v1=True, PairedBWAlignerV1 (the reference's default paired-end aligner)
        for one pair -> (hits: upstream mate, downstream mate, ..., n_hits_total, status)"""
        if getattr(self, "_refs", None) is None:
            self._refs = lib().bto_refs_build(C.byref(self.fw))
        hits = (OHit * cap)()
        tot, st = C.c_uint32(), C.c_uint32()
        fn = lib().bto_align_pair_v1 if v1 else lib().bto_align_pair_best
        n = fn(C.byref(self.fw), C.byref(self.bw) if self.bw is not None else None, self._refs,
                                      C.byref(pol), seq1.tobytes(), bytes(qual1), len(seq1), seed1,
                                      seq2.tobytes(), bytes(qual2), len(seq2), seed2, hits, cap,
                                      C.byref(tot), C.byref(st), C.byref(counts) if counts is not None else None)
        if n < 0:
            raise ValueError("oracle error %d" % -n)
        return self._unpack(hits, n), int(tot.value), int(st.value)


def rand_seed(seq: np.ndarray, qual: bytes, name: bytes, global_seed: int = 0) -> int:
    return int(lib().bto_rand_seed(seq.tobytes(), bytes(qual), len(seq), name, len(name), global_seed))
