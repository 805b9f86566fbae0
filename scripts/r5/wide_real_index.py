"""64-bit rows on a REAL index of more than 2^32 rows, on the CPU (there is no GPU budget left in round 5): the host build of
the wide automaton (tests/emu, -DBT_WIDE=1: libbowtie_amd_l.so's sources) against the unmodified reference's bowtie-align-l
on the same index and reads, SAM against SAM.

  genome : 5 sequences x 900 Mbp of seeded random bases (one 500-base N gap each), 4.5 x 10^9 rows  (/tmp/big64/gen.py)
  index  : the reference's own bowtie-build-l (oracle/_ref), 23 min per direction with --threads 5 here
  reads  : sampled from the genome, both strands, 0-3 substitutions, random qualities; bowtie-align-l -p 1 (with -p 4 --reorder the reference itself
           stopped making progress on one -a -m 20 run: 0.5 s of CPU in 5 minutes)

usage: wide_real_index.py <dir with genome.fa and g.*.ebwtl> <n reads> <read length> <mode> [<mode> ...]   (modes of tests/common.py)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common as T                                   # noqa: E402
import emu_lib as E                                  # noqa: E402
import refrun as R                                   # noqa: E402
from bowtie_amd import _abi as A                     # noqa: E402
from bowtie_amd.reads import Read, pack_reads        # noqa: E402
from bowtie_amd.synth import write_fastq             # noqa: E402

ARGS = {"v0": ["-v", "0"], "v2": ["-v", "2"], "n2": ["-n", "2"], "n2_k3": ["-n", "2", "-k", "3"], "v1": ["-v", "1"], "n3": ["-n", "3"],
        "n1_a_m20": ["-n", "1", "-a", "-m", "20"],
        # the best-first engine
        "n2_best": ["-n", "2", "--best"], "v3": ["-v", "3"], "n2_M3": ["-n", "2", "-M", "3"], "v2_a_best_strata": ["-v", "2", "-a", "--best", "--strata"],
        "n3_best": ["-n", "3", "--best"], "v2_best": ["-v", "2", "--best"], "n2_k2_best_strata_m5": ["-n", "2", "-k", "2", "--best", "--strata", "-m", "5"],
        # pairs (modes starting with pe: the reads are sampled as fragments of 200-450 bases, --fr)
        "pe_n1_best_X500": ["-n", "1", "--best", "-X", "500"], "pe_n2_best_X400_I250_k3": ["-n", "2", "--best", "-X", "400", "-I", "250", "-k", "3"],
        "pev1_n2_X500": ["-n", "2", "-X", "500"], "pev1_v2_X500": ["-v", "2", "-X", "500"]}


def sample_reads(fa, n, L, seed, pairs=False):
    mm = np.memmap(fa, dtype=np.uint8, mode="r")
    # sequence starts: '>' at the start of the file and after every 900 Mbp body (60 bases + newline per line)
    starts, pos = [], 0
    body = 900_000_000 // 60 * 61
    while pos < len(mm):
        assert mm[pos] == ord(">")
        e = pos
        while mm[e] != 10:
            e += 1
        starts.append(e + 1)
        pos = e + 1 + body
    rng = np.random.default_rng(seed)
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    def stretch(c, p, n_):
        o0 = starts[c] + (p // 60) * 61 + p % 60
        raw = np.array(mm[o0:o0 + n_ + n_ // 60 + 2])
        return lut[raw[raw != 10][:n_]]

    def mutate(seq):
        seq = seq.copy()
        for _ in range(int(rng.choice([0, 0, 1, 2, 2, 3]))):
            k = int(rng.integers(0, len(seq)))
            seq[k] = (seq[k] + int(rng.integers(1, 4))) & 3
        return seq, (rng.integers(10, 41, size=len(seq)) + 33).astype(np.uint8).tobytes()

    reads, mates = [], []
    while len(reads) < n:
        c = int(rng.integers(0, len(starts)))
        if pairs:
            F = int(rng.integers(200, 451))
            p = int(rng.integers(0, 900_000_000 - F))
            frag = stretch(c, p, F)
            if (frag == 4).any():
                continue
            if rng.integers(0, 2):
                frag = (3 - frag)[::-1]
            a, qa = mutate(frag[:L])
            b, qb = mutate((3 - frag[F - L:])[::-1])
            reads.append(Read(b"p%d/1" % len(reads), a, qa))
            mates.append(Read(b"p%d/2" % len(mates), b, qb))
            continue
        p = int(rng.integers(0, 900_000_000 - L))
        seq = stretch(c, p, L)
        if (seq == 4).any():
            continue
        if rng.integers(0, 2):
            seq = (3 - seq)[::-1]
        seq, qual = mutate(seq)
        reads.append(Read(b"r%d" % len(reads), seq, qual))
    return (pack_reads(reads), pack_reads(mates)) if pairs else pack_reads(reads)


def main():
    d, n, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    modes = sys.argv[4:]
    base = os.path.join(d, "g")
    batch = sample_reads(os.path.join(d, "genome.fa"), n, L, 20260927)
    fq = os.path.join(d, "reads_%d_%d.fq" % (n, L))
    write_fastq(batch, fq)
    b1 = b2 = None
    if any(m.startswith("pe") for m in modes):
        b1, b2 = sample_reads(os.path.join(d, "genome.fa"), n, L, 20260928, pairs=True)
        f1, f2 = os.path.join(d, "pairs_%d_%d_1.fq" % (n, L)), os.path.join(d, "pairs_%d_%d_2.fq" % (n, L))
        write_fastq(b1, f1)
        write_fastq(b2, f2)
    need_mirror = any(m != "v0" for m in modes)
    t0 = time.time()
    emu = E.EmuAligner(base, need_mirror=need_mirror, wide=True)
    ln, bias, width = emu.dims()
    print("index: %d rows (2^32 = %d), %d-byte rows, loaded by the wide host build in %.0f s" % (ln + 1, 1 << 32, width, time.time() - t0), flush=True)
    assert width == 8 and bias == 0 and ln + 1 > (1 << 32)
    # how much of the search runs above 2^32: LF of the last row, and of a row in the middle
    lf, _ = emu.rank4(ln)
    print("LF(last row) = %s" % lf, flush=True)
    refnames = ["chr%d" % (i + 1) for i in range(5)]
    ok = True
    for mode in modes:
        kw = T.MODES[mode]
        paired = mode.startswith("pe")
        t0 = time.time()
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bowtie-align-l"), "--wrapper", "basic-0", "-p", "1", "-S", "--sam-nohead", "-t"] + ARGS[mode]
                           + ([base, "-1", f1, "-2", f2] if paired else [base, fq]), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
        assert p.returncode == 0, p.stderr.decode()
        t_ref = time.time() - t0
        t0 = time.time()
        if paired:
            pol = A.make_policy(**(dict(kw, pe_v1=True) if mode.startswith("pev1") else kw))
            res = emu.align_pairs(pol, b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
            got = R.render_pairs(b1, b2, res, refnames, sam=True, mhits=kw.get("mhits", 0xFFFFFFFF))
        else:
            res = emu.align(A.make_policy(**kw), batch, hit_cap=T.hit_cap_for(kw), pal_cap=16384, n_lanes=64)
            got = R.render(batch, res, refnames, sam=True, mhits=kw.get("mhits", 0xFFFFFFFF), sample_max=kw.get("sample_max", False))
        t_emu = time.time() - t0
        want = p.stdout
        aligned = sum(1 for h, _, _ in res if h)
        high = sum(1 for ln_ in want.split(b"\n") if ln_ and not ln_.startswith(b"@"))
        same = got == want
        ok &= same
        print("%-9s %d reads x %d bp: %d aligned; SAM %s bowtie-align-l's (%d lines); reference %.0f s (with index load), wide host build %.0f s"
              % (mode, n, L, aligned, "IDENTICAL to" if same else "DIFFERS from", high, t_ref, t_emu), flush=True)
        if not same:
            g, w = T.strip_sam(got).split(b"\n"), T.strip_sam(want).split(b"\n")
            for i, (a, b) in enumerate(zip(g, w)):
                if a != b:
                    print("  first difference at line %d:\n   got  %r\n   want %r" % (i, a, b))
                    break
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
