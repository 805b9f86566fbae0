"""TEST INFRASTRUCTURE: a second, independent writer of the reference's default ("verbose") and SAM record formats, in
Python -- what the golden SAMs are compared through in the oracle / emulator / GPU parity tests, so that those do not
depend on the product's C++ formatter (which has its own tests against the reference, test_hostio.py).

Mirrors (behaviour, not code):
  * VerboseHitSink::append           hit.cpp:73-301
  * SAMHitSink::append               sam.cpp:129-257
  * SAMHitSink::reportUnOrMax        sam.cpp:57-124
  * SAMHitSink::appendHeaders        sam.cpp:20-49
  * HitSink::finish (stderr summary) hit.h:270-346
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from bowtie_amd.output import Hit
from bowtie_amd.reads import decode_seq

_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def _upto_ws(name: str) -> str:
    for i, ch in enumerate(name):
        if ch in " \t\n\r\x0b\x0c":
            return name[:i]
    return name


def _oriented(seq: np.ndarray, qual: bytes, fw: bool):
    """Hit::patSeq / Hit::quals: the read as aligned (rc'd / reversed for '-' hits)."""
    if fw:
        return seq, qual
    return _COMP[seq[::-1]], qual[::-1]


def format_verbose(name: bytes, seq: np.ndarray, qual: bytes, hit: Hit, refnames: Sequence[str],
                   full_ref: bool = False, off_base: int = 0) -> bytes:
    pseq, pqual = _oriented(seq, qual, hit.fw)
    L = len(seq)
    ref = refnames[hit.tidx] if hit.tidx < len(refnames) else str(hit.tidx)
    if not full_ref:
        ref = _upto_ws(ref)
    mm = []
    for pos, refc in hit.mms:
        qry = pseq[pos] if hit.fw else pseq[L - pos - 1]
        mm.append("%d:%s>%s" % (pos, "ACGT"[refc], "ACGTN"[qry]))
    parts = [name, b"+" if hit.fw else b"-", ref.encode(), str(hit.toff + off_base).encode(),
             decode_seq(pseq), pqual, str(hit.oms).encode(), ",".join(mm).encode()]
    return b"\t".join(parts) + b"\n"


def _qname(name: bytes, trunc: bool = True) -> bytes:
    if not trunc:
        return name
    for i, ch in enumerate(name):
        if ch in b" \t\n\r\x0b\x0c":
            return name[:i]
    return name


def format_sam(name: bytes, seq: np.ndarray, qual: bytes, hit: Hit, refnames: Sequence[str],
               mapq: int = 255, xms: int = 0, full_ref: bool = False, mate_hit: "Hit | None" = None,
               mate_len: int = 0) -> bytes:
    """mate_hit / mate_len: the partner alignment of a paired hit (sam.cpp:129-257: flags 1|2|64/128|32,
    MRNM '=', MPOS, ISIZE; QNAME loses its /1 or /2)."""
    pseq, pqual = _oriented(seq, qual, hit.fw)
    L = len(seq)
    ref = refnames[hit.tidx] if hit.tidx < len(refnames) else str(hit.tidx)
    if not full_ref:
        ref = _upto_ws(ref)
    flags = 0 if hit.fw else 16
    mrnm, mpos, isize = b"*", b"0", b"0"
    if mate_hit is not None:
        flags |= 1 | 2 | (64 if hit.mate == 1 else 128) | (0 if mate_hit.fw else 32)
        mrnm, mpos = b"=", str(mate_hit.toff + 1).encode()
        ins = -(hit.toff - mate_hit.toff + L) if hit.toff > mate_hit.toff else (mate_hit.toff - hit.toff + mate_len)
        isize = str(ins).encode()
        name = name[:-2] if len(name) >= 2 else b""
    # MD:Z walks the alignment left to right on the reference
    mmd = {pos: refc for pos, refc in hit.mms}
    order = range(L) if hit.fw else range(L - 1, -1, -1)
    md = []
    run = 0
    nm = 0
    for i in order:
        if i in mmd:
            nm += 1
            md.append("%d%s" % (run, "ACGT"[mmd[i]]))
            run = 0
        else:
            run += 1
    md.append(str(run))
    out = [_qname(name), str(flags).encode(), ref.encode(), str(hit.toff + 1).encode(),
           str(mapq).encode(), ("%dM" % L).encode(), mrnm, mpos, isize, decode_seq(pseq), pqual,
           ("XA:i:%d" % hit.stratum).encode(), ("MD:Z:" + "".join(md)).encode(),
           ("NM:i:%d" % nm).encode()]
    if xms > 0:
        out.append(("XM:i:%d" % xms).encode())
    return b"\t".join(out) + b"\n"


def format_sam_unaligned(name: bytes, seq: np.ndarray, qual: bytes, n_maxed_hits: int = 0, mate: int = 0) -> bytes:
    flag = b"4" if mate == 0 else (b"77" if mate == 1 else b"141")
    if mate:
        name = name[:-2] if len(name) >= 2 else b""
    out = [_qname(name), flag, b"*", b"0", b"0", b"*", b"*", b"0", b"0", decode_seq(seq), qual,
           ("XM:i:%d" % n_maxed_hits).encode()]
    return b"\t".join(out) + b"\n"


def sam_header(refnames: Sequence[str], reflens: Sequence[int], cmdline: str,
               full_ref: bool = False, version: str = "1.3.1") -> bytes:
    o = ["@HD\tVN:1.0\tSO:unsorted\n"]
    for nm, ln in zip(refnames, reflens):
        o.append("@SQ\tSN:%s\tLN:%d\n" % (nm if full_ref else _upto_ws(nm), ln))
    o.append('@PG\tID:Bowtie\tVN:%s\tCL:"%s"\n' % (version, cmdline))
    return "".join(o).encode()


def summary(n_aligned: int, n_unaligned: int, n_maxed: int, n_reported: int) -> str:
    """HitSink::finish stderr summary (hit.h:279-337), unpaired, no -M."""
    tot = n_aligned + n_unaligned + n_maxed
    al = 100.0 * (n_aligned + n_maxed) / tot if tot else 0.0
    un = 100.0 * n_unaligned / tot if tot else 0.0
    mx = 100.0 * n_maxed / tot if tot else 0.0
    s = "# reads processed: %d\n" % tot
    s += "# reads with at least one alignment: %d (%.2f%%)\n" % (n_aligned + n_maxed, al)
    s += "# reads that failed to align: %d (%.2f%%)\n" % (n_unaligned, un)
    if n_maxed > 0:
        s += "# reads with alignments suppressed due to -m: %d (%.2f%%)\n" % (n_maxed, mx)
    s += ("No alignments\n" if n_reported == 0 else "Reported %d alignments\n" % n_reported)
    return s
