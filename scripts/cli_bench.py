#!/usr/bin/env python3
"""End-to-end rate of the bowtie-amd binary (FASTQ file in -> SAM file out, PCIe and host parsing /
formatting included) next to the unmodified reference binary on the same files and host.
    python scripts/cli_bench.py [--index ecoli|big] [--reads N] [--len L] [--threads T]
Run on the GPU box; prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                    # noqa: E402
from bowtie_amd import aligner as AL                  # noqa: E402
from bowtie_amd.synth import synth_reads, write_fastq  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", default="ecoli")
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--mode", default="-n 2")
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--extra", default="", help="extra bowtie-amd options")
    a = ap.parse_args()
    if a.index == "ecoli":
        base = os.path.join(ROOT, "tests", "golden", "e_coli")
        text = AL.restore_text(base)
    else:
        import torch
        from bowtie_amd import ebwt_build as EB
        base, text, _ = EB.ensure_big_index(0, torch.device("cuda", 0))
    fq = "/tmp/cli_bench_%s_%d.fq" % (a.index, a.reads)
    if not os.path.exists(fq):
        write_fastq(synth_reads(text, a.reads, a.len, mm_dist=(0, 1, 2, 2, 3, 4), seed=99), fq)
    mode = a.mode.split()
    out = {"index": a.index, "reads": a.reads, "len": a.len, "mode": a.mode, "fastq_bytes": os.path.getsize(fq)}
    ours = [os.path.join(ROOT, "bowtie_amd", "bowtie-amd"), "-p", str(a.threads), "-t", "-S"] + a.extra.split() + mode + ["-x", base, fq, "/tmp/cli_ours.sam"]
    t0 = time.perf_counter()
    p = subprocess.run(ours, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out["bowtie_amd_s"] = time.perf_counter() - t0
    out["bowtie_amd_reads_per_s"] = a.reads / out["bowtie_amd_s"]
    lines = p.stderr.decode(errors="replace").strip().split("\n")
    out["bowtie_amd_stderr"] = lines if os.environ.get("BT_CLI_TIMELINE") else lines[-8:]       # the timeline comes after the summary
    ref = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")
    if os.path.exists(ref) and not a.no_ref:
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        p = subprocess.run([ref, "--wrapper", "basic-0", "-p", str(cores), "-t", "-S"] + mode + ["-x", base, fq, "/tmp/cli_ref.sam"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out["reference_s"] = time.perf_counter() - t0
        out["reference_cores"] = cores
        out["reference_reads_per_s"] = a.reads / out["reference_s"]
        # same alignments: the reference with -p N writes reads in a different order
        def digest(path):
            import hashlib
            h = 0
            with open(path, "rb") as f:
                for line in f:
                    if not line.startswith(b"@"):
                        h ^= int.from_bytes(hashlib.md5(line).digest()[:8], "little")
            return h
        out["same_alignment_multiset"] = digest("/tmp/cli_ours.sam") == digest("/tmp/cli_ref.sam")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
