#!/usr/bin/env python3
"""The FASTQ reader alone (bt_reads_open / bt_reads_next through the C ABI; no GPU, no search, no output): reads per second
of the input stage at steady state -- batch after batch of one file that sits in the page cache, the first batch (fresh
memory: page faults) left out of the rate.
    python scripts/r6/reader_bench.py [--reads N] [--batch B] [--threads T] [--len L]
Prints one JSON line; BT_IO_PROFILE=1 adds the reader's own phase times per batch on stderr."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bowtie_amd import _abi as A          # noqa: E402
from bowtie_amd.hostio import lib, FORMATS, QUALS   # noqa: E402
from cli_host_bench import write_fastq_fast          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--gz", action="store_true")
    a = ap.parse_args()
    fq = "/tmp/cli_host_bench_%d_%d.fq" % (a.reads, a.len)
    if not os.path.exists(fq):
        write_fastq_fast(fq, a.reads, a.len)
    if a.gz and not os.path.exists(fq + ".gz"):
        os.system("gzip -1 -k %s" % fq)
    path = fq + (".gz" if a.gz else "")
    with open(path, "rb") as f:
        while f.read(1 << 26):
            pass
    L = lib()
    o = A.ReadOpts(FORMATS["fastq"], 0, 0, QUALS["phred33"], 0, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert L.bt_reads_open(path.encode(), C.byref(o), C.byref(h)) == 0
    times, total, digest = [], 0, 0
    while True:
        rb = A.ReadBatchC()
        names, noff = C.c_void_p(), C.c_void_p()
        t0 = time.perf_counter()
        rc = L.bt_reads_next(h, a.batch, a.threads, C.byref(rb), C.byref(names), C.byref(noff))
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        if rb.n_reads == 0:
            break
        times.append((rb.n_reads, dt))
        total += rb.n_reads
    L.bt_reads_close(h)
    steady = times[1:] if len(times) > 2 else times
    n_s, t_s = sum(n for n, _ in steady), sum(t for _, t in steady)
    print(json.dumps({"file": os.path.basename(path), "bytes": os.path.getsize(path), "reads": total, "batch": a.batch, "threads": a.threads,
                      "first_batch_s": times[0][1], "steady_reads_per_s": n_s / t_s, "steady_GB_per_s": os.path.getsize(path) * (n_s / total) / t_s / 1e9,
                      "batches": len(times)}))


if __name__ == "__main__":
    main()
