#!/usr/bin/env python3
"""Ceiling of the bowtie-amd binary's host pipeline: FASTQ file in -> SAM file out with the search taken away.  The binary runs
under tests/emu/libcli_shim.so (LD_PRELOAD, test infrastructure) in its SHIM_NULL_SEARCH mode -- every batch is answered
at once with made-up alignments for three reads in four -- so the wall time is the reader, the batching, the formatter and
the writer on this host's cores.  No GPU.
    python scripts/cli_host_bench.py [--reads N] [--len L] [--threads T] [--extra "..."]
Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                    # noqa: E402


def write_fastq_fast(path, n, L, seed=5):
    """n records of L bases: record i is '@r<i>\\n<bases>\\n+\\n<quals>\\n' (numpy, no per-read Python)."""
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        done = 0
        while done < n:
            m = min(1 << 20, n - done)
            names = np.char.add("@r", np.arange(done, done + m).astype("U12")).astype("S")
            bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, L))]
            quals = (rng.integers(20, 41, size=(m, L)) + 33).astype(np.uint8)
            rows = [names[i] + b"\n" + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(m)]
            f.write(b"".join(rows))
            done += m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--extra", default="")
    ap.add_argument("--out", default="/tmp/cli_host_bench.sam")
    a = ap.parse_args()
    import emu_lib
    shim = emu_lib.shim()
    base = os.path.join(ROOT, "tests", "golden", "e_coli")
    fq = "/tmp/cli_host_bench_%d_%d.fq" % (a.reads, a.len)
    if not os.path.exists(fq):
        write_fastq_fast(fq, a.reads, a.len)
    with open(fq, "rb") as f:                         # into the page cache
        while f.read(1 << 26):
            pass
    cmd = [os.path.join(ROOT, "bowtie_amd", "bowtie-amd"), "-p", str(a.threads), "-t", "-S"] + a.extra.split() + ["-x", base, fq, a.out]
    env = dict(os.environ, LD_PRELOAD=shim, SHIM_NULL_SEARCH="1")
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    dt = time.perf_counter() - t0
    out = {"reads": a.reads, "len": a.len, "threads": a.threads, "fastq_bytes": os.path.getsize(fq),
           "sam_bytes": os.path.getsize(a.out) if os.path.exists(a.out) else 0, "rc": p.returncode,
           "wall_s": dt, "host_pipeline_reads_per_s": a.reads / dt,
           "stderr": p.stderr.decode(errors="replace").strip().split("\n")[-16:]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
