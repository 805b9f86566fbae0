"""Helpers shared by tests and oracle/gen_golden.py: run a search through a backend and render the
reference's output formats; run the unmodified reference binary (oracle/_ref) when present."""
from __future__ import annotations

import os
import subprocess
from typing import List, Optional

import pyformat as O
from bowtie_amd.reads import ReadBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")


def have_ref_binary() -> bool:
    return os.path.exists(REF_BIN)


def run_reference(args: List[str], index_base: str, reads_path: str) -> bytes:
    """stdout of the unmodified reference, -p 1 (deterministic order)."""
    cmd = [REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", index_base, reads_path]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return p.stdout


def _rand_u32(seed: int) -> int:
    """First RandomSource::nextU32 after init(seed) (random_source.h:45-54)."""
    last = (1664525 * seed + 1013904223) & 0xFFFFFFFF
    ret = last >> 16
    last = (1664525 * last + 1013904223) & 0xFFFFFFFF
    return ret ^ last


def render(batch: ReadBatch, per_read, refnames, sam: bool, mhits: int = 0xFFFFFFFF,
           sample_max: bool = False) -> bytes:
    """per_read[i] = (hits(list of output.Hit), n_hits_total, status) -> reference-format text.

    Verbose mode prints nothing for unaligned/maxed reads (hit.h:494-500); SAM prints a flag-4
    record (sam.cpp:57-124)."""
    out = []
    for i in range(batch.n):
        hits, total, status = per_read[i]
        L = int(batch.len[i])
        seq = batch.seq[i, :L]
        qual = batch.qual[i, :L].tobytes()
        name = batch.names[i]
        maxed = total > mhits
        if hits and not maxed:
            for h in hits:
                if sam:
                    out.append(O.format_sam(name, seq, qual, h, refnames, xms=len(hits)))
                else:
                    out.append(O.format_verbose(name, seq, qual, h, refnames))
        elif maxed and sample_max and hits:
            # -M: one of the hits tied for the best stratum, picked with the read's seed
            # (VerboseHitSink::reportMaxed hit.cpp:16-68, SAMHitSink::reportMaxed sam.cpp:263-311)
            num = 1
            for k in range(1, len(hits)):
                if hits[k].stratum == hits[k - 1].stratum:
                    num += 1
                else:
                    break
            h = hits[_rand_u32(int(batch.seed[i])) % num]
            if sam:
                out.append(O.format_sam(name, seq, qual, h, refnames, mapq=0, xms=len(hits) + 1))
            else:
                import dataclasses
                out.append(O.format_verbose(name, seq, qual, dataclasses.replace(h, oms=len(hits)), refnames))
        elif sam and not maxed:
            # -m-suppressed reads print nothing unless -M (hit.h:494-500, sam.cpp:263-269)
            out.append(O.format_sam_unaligned(name, seq, qual, 0))
    return b"".join(out)


def oracle_search(oidx, pol, batch: ReadBatch, cap: Optional[int] = None, counts=None):
    cap = cap or (64 if pol.all_hits else max(1, min(int(pol.khits), 64)))
    res = []
    for i in range(batch.n):
        L = int(batch.len[i])
        hits, total, st = oidx.align(pol, batch.seq[i, :L], batch.qual[i, :L].tobytes(),
                                     int(batch.seed[i]), cap=cap, counts=counts)
        res.append(([O.Hit(h["tidx"], h["toff"], h["oms"], h["cost"], h["stratum"], h["fw"], h["mms"])
                     for h in hits], total, st))
    return res


def run_reference_pairs(args: List[str], index_base: str, reads1: str, reads2: str) -> bytes:
    cmd = [REF_BIN, "--wrapper", "basic-0", "-p", "1"] + args + ["-x", index_base, "-1", reads1, "-2", reads2]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return p.stdout


def oracle_search_pairs(oidx, pol, b1: ReadBatch, b2: ReadBatch, cap: Optional[int] = None, counts=None, v1: bool = False):
    from bowtie_amd.aligner import pair_hit_cap
    cap = cap or pair_hit_cap(pol)
    res = []
    for i in range(b1.n):
        L1, L2 = int(b1.len[i]), int(b2.len[i])
        hits, total, st = oidx.align_pair(pol, b1.seq[i, :L1], b1.qual[i, :L1].tobytes(), int(b1.seed[i]),
                                          b2.seq[i, :L2], b2.qual[i, :L2].tobytes(), int(b2.seed[i]),
                                          cap=cap, counts=counts, v1=v1)
        res.append(([O.Hit(h["tidx"], h["toff"], h["oms"], h["cost"], h["stratum"], h["fw"], h["mms"], h["mate"])
                     for h in hits], total, st))
    return res


def render_pairs(b1: ReadBatch, b2: ReadBatch, per_pair, refnames, sam: bool, mhits: int = 0xFFFFFFFF,
                 sample_max: bool = False) -> bytes:
    """per_pair[i] = (hits: upstream mate, downstream mate, ..., n_hits_total, status) -> reference text.
    finishRead (hit.h:741-786) with the doubled -k/-m of createMult(2); XM:i = pairs reported."""
    out = []
    maxv = 0xFFFFFFFF if mhits == 0xFFFFFFFF else 2 * mhits
    for i in range(b1.n):
        hits, total, status = per_pair[i]
        reads = {1: (b1.names[i], b1.seq[i, :int(b1.len[i])], b1.qual[i, :int(b1.len[i])].tobytes()),
                 2: (b2.names[i], b2.seq[i, :int(b2.len[i])], b2.qual[i, :int(b2.len[i])].tobytes())}
        maxed = total > maxv
        if hits and not maxed:
            for k in range(0, len(hits) - 1, 2):
                for h, m in ((hits[k], hits[k + 1]), (hits[k + 1], hits[k])):
                    name, seq, qual = reads[h.mate]
                    if sam:
                        out.append(O.format_sam(name, seq, qual, h, refnames, xms=len(hits) // 2, mate_hit=m,
                                                mate_len=len(reads[m.mate][1])))
                    else:
                        out.append(O.format_verbose(name, seq, qual, h, refnames))
        elif maxed and sample_max and len(hits) >= 2:
            # -M for pairs (hit.cpp:27-55, sam.cpp:274-299): of the buffered pairs, those whose better mate is in the best
            # stratum; one of them, picked with the first draw of the first mate's generator
            strata = [min(hits[k].stratum, hits[k + 1].stratum) for k in range(0, len(hits) - 1, 2)]
            best = min(strata)
            cands = [k for k, st in enumerate(strata) if st == best]
            k = 2 * cands[_rand_u32(int(b1.seed[i])) % len(cands)]
            for h, m in ((hits[k], hits[k + 1]), (hits[k + 1], hits[k])):
                name, seq, qual = reads[h.mate]
                if sam:
                    out.append(O.format_sam(name, seq, qual, h, refnames, mapq=0, xms=len(hits) // 2 + 1, mate_hit=m,
                                            mate_len=len(reads[m.mate][1])))
                else:
                    import dataclasses
                    out.append(O.format_verbose(name, seq, qual, dataclasses.replace(h, oms=len(hits) // 2), refnames))
        elif sam and not maxed:
            for mate in (1, 2):
                name, seq, qual = reads[mate]
                out.append(O.format_sam_unaligned(name, seq, qual, 0, mate=mate))
    return b"".join(out)
