"""Re-writes of a small `.ebwt` index in the other members of the family (test infrastructure): the
same index other-endian, and in bowtie2-build's side layout (`.bt2`).  oracle/gen_golden_family.py
feeds them to the unmodified reference binary and records that its output does not change; the
tests feed them to the product's loader.  Format: ebwt.h:138-183 (EbwtParams::init), 2926-3272
(readIntoMemory), 2240-2328 (countBt2Side[Ex]); reference.h:35-120 (.3/.4)."""
from __future__ import annotations

import os
import shutil
import struct

import numpy as np

EXTS = ("1", "2", "rev.1", "rev.2", "3", "4")


def _parse1(path):
    b = open(path, "rb").read()
    one, ln, line_rate, lps, off_rate, ftab_chars, flags = struct.unpack_from("<iIiiiii", b, 0)
    assert one == 1 and line_rate == 6 and lps == 1
    p = 28
    n_pat, = struct.unpack_from("<I", b, p); p += 4
    plen = np.frombuffer(b, "<u4", n_pat, p); p += 4 * n_pat
    n_frag, = struct.unpack_from("<I", b, p); p += 4
    rstarts = np.frombuffer(b, "<u4", 3 * n_frag, p); p += 12 * n_frag
    bwt_sz = ln // 4 + 1
    tot = (bwt_sz + 111) // 112 * 128
    ebwt = np.frombuffer(b, np.uint8, tot, p); p += tot
    z_off, = struct.unpack_from("<I", b, p); p += 4
    fchr = np.frombuffer(b, "<u4", 5, p); p += 20
    ftab_len = (1 << (2 * ftab_chars)) + 1
    ftab = np.frombuffer(b, "<u4", ftab_len, p); p += 4 * ftab_len
    eftab = np.frombuffer(b, "<u4", 2 * ftab_chars, p); p += 8 * ftab_chars
    return dict(len=ln, line_rate=line_rate, lps=lps, off_rate=off_rate, ftab_chars=ftab_chars, flags=flags,
                plen=plen, rstarts=rstarts, ebwt=ebwt, z_off=z_off, fchr=fchr, ftab=ftab, eftab=eftab, names=b[p:])


def _write1(path, h, ebwt: bytes, end: str, line_rate=None, lps=None):
    with open(path, "wb") as f:
        f.write(struct.pack(end + "iIiiiii", 1, h["len"], line_rate or h["line_rate"], lps or h["lps"], h["off_rate"],
                            h["ftab_chars"], h["flags"]))
        f.write(struct.pack(end + "I", len(h["plen"])))
        f.write(h["plen"].astype(end + "u4").tobytes())
        f.write(struct.pack(end + "I", len(h["rstarts"]) // 3))
        f.write(h["rstarts"].astype(end + "u4").tobytes())
        f.write(ebwt)
        f.write(struct.pack(end + "I", h["z_off"]))
        f.write(h["fchr"].astype(end + "u4").tobytes())
        f.write(h["ftab"].astype(end + "u4").tobytes())
        f.write(h["eftab"].astype(end + "u4").tobytes())
        f.write(h["names"])


def _swap_words(path_in, path_out, end):
    b = open(path_in, "rb").read()
    assert len(b) % 4 == 0
    with open(path_out, "wb") as f:
        f.write(np.frombuffer(b, "<u4").astype(end + "u4").tobytes())


def _swap_side_counters(ebwt: np.ndarray) -> bytes:
    e = ebwt.reshape(-1, 64).copy()
    e[:, 56:64] = e[:, 56:64].reshape(-1, 2, 4)[:, :, ::-1].reshape(-1, 8)      # the two u32 counters of each side
    return e.tobytes()


def _swap3(path_in, path_out):
    b = open(path_in, "rb").read()
    one, n = struct.unpack_from("<II", b, 0)
    assert one == 1
    with open(path_out, "wb") as f:
        f.write(struct.pack(">II", 1, n))
        for i in range(n):
            off, ln, first = struct.unpack_from("<IIB", b, 8 + 9 * i)
            f.write(struct.pack(">IIB", off, ln, first))


def write_swapped(src: str, dst: str):
    """<src>.*.ebwt -> <dst>.*.ebwt as a big-endian machine's bowtie-build would have written them."""
    for rev in ("", ".rev"):
        h = _parse1(src + rev + ".1.ebwt")
        _write1(dst + rev + ".1.ebwt", h, _swap_side_counters(h["ebwt"]), ">")
        _swap_words(src + rev + ".2.ebwt", dst + rev + ".2.ebwt", ">")
    if os.path.exists(src + ".3.ebwt"):
        _swap3(src + ".3.ebwt", dst + ".3.ebwt")
        shutil.copyfile(src + ".4.ebwt", dst + ".4.ebwt")


def bwt_symbols(h) -> np.ndarray:
    """The BWT column of a parsed small index, one symbol per row (the '$' row holds an A), whole sides."""
    e = h["ebwt"].reshape(-1, 64)[:, :56]
    sym = np.stack([(e >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(len(e), 224)
    sym[0::2] = sym[0::2, ::-1]                                   # backward sides run the other way
    return sym.reshape(-1)


def write_bt2(src: str, dst: str):
    """<src>.*.ebwt -> <dst>.*.bt2: all sides forward, 48 BWT bytes + [A][C][G][T] counts of the rows before
    the side ('$' not counted)."""
    for rev in ("", ".rev"):
        h = _parse1(src + rev + ".1.ebwt")
        ln = h["len"]
        bwt = bwt_symbols(h)[:ln + 1]
        bwt_sz = ln // 4 + 1
        n_sides = (bwt_sz + 47) // 48
        pad = np.zeros(n_sides * 192, dtype=np.uint8)
        pad[:ln + 1] = bwt
        sides = pad.reshape(n_sides, 192)
        counted = np.ones(n_sides * 192, dtype=bool)
        counted[h["z_off"]] = False
        occ = np.stack([((sides == c) & counted.reshape(n_sides, 192)).sum(1) for c in range(4)], 1)
        before = np.zeros_like(occ)
        before[1:] = np.cumsum(occ, 0)[:-1]
        q = sides.reshape(n_sides, 48, 4)
        out = np.zeros((n_sides, 64), dtype=np.uint8)
        out[:, :48] = q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)
        out[:, 48:] = before.astype("<u4").view(np.uint8).reshape(n_sides, 16)
        _write1(dst + rev + ".1.bt2", h, out.tobytes(), "<", lps=2)
        shutil.copyfile(src + rev + ".2.ebwt", dst + rev + ".2.bt2")
    if os.path.exists(src + ".3.ebwt"):
        shutil.copyfile(src + ".3.ebwt", dst + ".3.bt2")
        shutil.copyfile(src + ".4.ebwt", dst + ".4.bt2")
