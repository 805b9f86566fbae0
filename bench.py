#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the FM-index search hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--reads R]

A "step" is one pass of the hot path (bt_align_batch_device: one persistent-lane search kernel)
over one batch of synthetic reads that already sits in HBM.  One process per GPU (torchrun);
reads are sharded by rank, the index is replicated in every GPU's HBM, and the only collective
is the final all-reduce of the hit counters (RCCL over xGMI).  Rank 0 prints one JSON line.

Workloads (BASELINE.json configs):
    ecoli_v0_36    e_coli index, 36-bp reads, -v 0              (config 2)
    ecoli_v2_76    e_coli index, 76-bp reads, -v 2
    ecoli_n2_100   e_coli index, 100-bp reads, -n 2 -l 28 -e 70
    big_v2_76      hg19-scale synthetic genome, 50 M x 76-bp, -v 2      (config 3)
    big_n2_100     hg19-scale synthetic genome, 200 M x 100-bp, -n 2 -l 28 per GPU per step
                   (config 4; the headline metric, default)
The hg19-scale index is synthesised on the GPU at start-up (bowtie_amd/ebwt_build.py): neither
hg19 nor any network exists on the bench box.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from bowtie_amd import _abi as A            # noqa: E402
from bowtie_amd import aligner as AL        # noqa: E402
from bowtie_amd.synth import synth_reads_torch, synth_pairs_torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
# HBM-side traffic per read, measured with rocprofv3 PMC (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE
# passes, gfx950-corrected as MI355X_MICROARCH.md prescribes): bench.py cannot run the profiler on itself,
# so the figures live in profiles/traffic.json keyed by (kernel template instance, workload) and are used
# only when the kernel this run launched is the one that was profiled -- otherwise `traffic` is null.
def gather_ceiling(index_kind: str, al=None, cus: int = 0):
    """Measured ceiling of the access pattern itself on this GPU: independent random rank queries over the index image
    with nothing else going on (bt_bench_gather; the 32-byte rank blocks the kernels gather), best configuration,
    priced at SURVEY 8(d)'s 128 bytes per query so that it compares with roofline.achieved.  Measured in this run when a
    context is at hand (< 1 s), otherwise read from an earlier round's file."""
    if al is not None and cus > 0:
        try:
            lib = AL.lib()
            lib.bt_bench_gather.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
            best = None
            for bpc in (2, 4, 8):
                ms, gbs = C.c_float(), C.c_double()
                nb, iters = cus * bpc, 2048
                if lib.bt_bench_gather(al._h, 0, nb, iters, 0, C.byref(ms), C.byref(gbs)) != 0:
                    raise RuntimeError("bt_bench_gather")
                q = nb * 256.0 * iters / (ms.value * 1e-3)
                if best is None or q > best[0]:
                    best = (q, bpc)
            return {"GBps": best[0] * 128.0 / 1e9, "Gqueries_per_s": best[0] / 1e9,
                    "source": "measured in this run: bt_bench_gather, %d blocks of 256 lanes per CU, 2048 independent rank queries per lane" % best[1]}
        except Exception as e:      # noqa: BLE001  -- the file of an earlier round stands in
            log("[bench] gather ceiling not measured (%s): reading an earlier round's" % e)
    src = ("profiles/r3/gather_big.json" if index_kind == "big" else "profiles/r2/gather_ecoli.json")
    try:
        with open(os.path.join(ROOT, src)) as f:
            j = json.load(f)
        best = max(j["runs"], key=lambda r: r["GBps_128B_per_query"])
        return {"GBps": best["GBps_128B_per_query"], "Gqueries_per_s": best["Gqueries_per_s"], "source": src}
    except (OSError, ValueError, KeyError):
        return None


# A PMC profile describes a kernel only as long as the kernel's source is what was profiled: the template name survives source
# changes, so an entry of profiles/traffic.json carries the SHA-256 (16 hex digits) of the source files its kernel is compiled
# from, and is used only while they are unchanged (round 5's hand-kept round numbers went stale: VERDICT r5).  Otherwise
# `traffic` is null.  `python bench.py --kernel-sha` prints the current digests.
KERNEL_SOURCES = {
    "bt_search_kernel": ("bt_rank.h", "bt_core.h", "bt_kernels.h", "bt_kernels.hip"),
    "bt_best_kernel": ("bt_rank.h", "bt_best.h", "bt_kernels.h", "bt_best_kernels.hip"),
    "bt_best_nested_kernel": ("bt_rank.h", "bt_best.h", "bt_kernels.h", "bt_best_kernels.hip"),
}


def kernel_source_sha16(kernel: str):
    import hashlib
    files = KERNEL_SOURCES.get(kernel.split("<")[0].split(" ")[0])
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "bowtie_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel: str, workload: str):
    try:
        sha = kernel_source_sha16(kernel)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if e["kernel"] == kernel and e["workload"] == workload and sha and e.get("source_sha16") == sha:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


SECTIONS = {}       # filled by the profiling build of bt_best_kernel (bt_best_prof_read)

WORKLOADS = {
    "ecoli_v0_36": dict(index="ecoli", length=36, pol=dict(mode="v", mms=0), mm_dist=(0,), reads=4_000_000),
    "ecoli_v2_76": dict(index="ecoli", length=76, pol=dict(mode="v", mms=2), mm_dist=(0, 0, 1, 1, 2, 3), reads=2_000_000),
    "ecoli_n2_100": dict(index="ecoli", length=100, pol=dict(mode="n", mms=2), mm_dist=(0, 1, 2, 2, 3, 4), reads=2_000_000),
    "big_v2_76": dict(index="big", length=76, pol=dict(mode="v", mms=2), mm_dist=(0, 0, 1, 1, 2, 3), reads=50_000_000),
    "big_n2_100": dict(index="big", length=100, pol=dict(mode="n", mms=2), mm_dist=(0, 1, 2, 2, 3, 4), reads=200_000_000),
    # the best-first engine (--best): bt_best_kernel
    "ecoli_n2_best_100": dict(index="ecoli", length=100, pol=dict(mode="n", mms=2, best=True), mm_dist=(0, 1, 2, 2, 3, 4), reads=2_000_000),
    "big_n2_best_100": dict(index="big", length=100, pol=dict(mode="n", mms=2, best=True), mm_dist=(0, 1, 2, 2, 3, 4), reads=32_000_000),
    # paired-end (BASELINE config 5): 2 x 50 bp, -n 1 --best -X 500; `reads` = pairs per GPU per step
    # (config 5 = 100 M pairs over 8 GPUs = 12.5 M per GPU)
    "ecoli_pe_n1_best_50": dict(index="ecoli", length=50, paired=True, pol=dict(mode="n", mms=1, best=True, max_ins=500),
                                mm_dist=(0, 0, 1, 1, 2), reads=1_000_000),
    "big_pe_n1_best_50": dict(index="big", length=50, paired=True, pol=dict(mode="n", mms=1, best=True, max_ins=500),
                              mm_dist=(0, 0, 1, 1, 2), reads=12_500_000),
    # the same pairs WITHOUT --best: the reference's default paired-end aligner, PairedBWAlignerV1 (aligner.h:606-1480)
    "big_pe_n1_50_v1": dict(index="big", length=50, paired=True, pol=dict(mode="n", mms=1, max_ins=500, pe_v1=True),
                            mm_dist=(0, 0, 1, 1, 2), reads=12_500_000),
    "ecoli_pe_n1_50_v1": dict(index="ecoli", length=50, paired=True, pol=dict(mode="n", mms=1, max_ins=500, pe_v1=True),
                              mm_dist=(0, 0, 1, 1, 2), reads=1_000_000),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(c: dict, n_reads: int, length: int, hits: int) -> float:
    """128 B per rank locus (one side pair), 4 B per ftab/offs word, 12 B per
    rstarts probe, read in (len x (seq+qual) + 16 B meta), hit out (32 B + 2 B/mm ~ 36 B)."""
    loci = 2 * (c["lfex"] + c["lf2"]) - c["same_pair"] + c["lf1"] + c["chase"]
    return 128.0 * loci + 4.0 * (2 * c["ftab"] + c["offs"]) + 12.0 * c["rstarts"] + \
        n_reads * (2.0 * length + 16.0) + hits * 36.0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _gpu_sam(idx, pol, batches):
    """SAM text of the product for a host-side sample (one ReadBatch, or two for pairs): bt_align_batch /
    bt_align_pairs + the C++ formatter -- what bowtie-amd prints for these reads."""
    from bowtie_amd import hostio as H
    lib = AL.lib()
    al = AL.Aligner(idx, pol)
    n = batches[0].n
    paired = len(batches) == 2
    cap = 2 if paired else 1
    keep = [AL.pack_batch(b) for b in batches]
    hits = np.zeros(n * cap, dtype=A.HIT_DTYPE)
    n_hits = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint8)
    pool = np.zeros(n * cap * 12 + 1024, dtype=np.uint16)
    hb = A.HitBatchC(cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
    if paired:
        if lib.bt_index_load_reference(idx._h) != 0:
            raise RuntimeError("bt_index_load_reference failed")
        rc = lib.bt_align_pairs(al._h, C.byref(keep[0][1]), C.byref(keep[1][1]), C.byref(hb), None)
    else:
        rc = lib.bt_align_batch(al._h, C.byref(keep[0][1]), C.byref(hb), None)
    if rc != 0:
        raise RuntimeError("GPU search of the diff sample failed: " + AL.strerror(rc))
    opts = H.out_opts(sam=True, khits=1)
    if paired:
        text, _ = H.format_pairs(batches[0], batches[1], hits, n_hits, status, pool, cap, idx.refnames, idx.reflens, opts)
    else:
        text, _ = H.format_hits(batches[0], hits, n_hits, status, pool, cap, idx.refnames, idx.reflens, opts)
    al.close()
    return text


def cpu_baseline(base: str, wl: dict, text_np: np.ndarray, idx=None, seconds: float = 10.0, diff_only: bool = False):
    """Reference bowtie (oracle/_ref, unmodified) on bounded FASTQ samples of the same workload, on this
    box's host cores: a -p sweep (two sample sizes per setting: the difference cancels index load and
    thread start-up), -p 1, the reference's own "Time searching" (-t), and -- parity at this scale,
    against the real thing -- the reference's SAM for the first sample compared line by line with the
    product's SAM for the same reads.  Falls back to the single-core C restatement when the binary is
    not on the box."""
    import re
    import subprocess
    import tempfile
    from bowtie_amd.synth import synth_reads, synth_pairs, write_fastq
    pol = wl["pol"]
    args = ["-v", str(pol["mms"])] if pol["mode"] == "v" else ["-n", str(pol["mms"]), "-l", "28", "-e", "70"]
    if pol.get("best"):
        args.append("--best")
    paired = bool(wl.get("paired"))
    if paired:
        args += ["-X", str(pol.get("max_ins", 250))]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")
    cores = os.cpu_count() or 1
    unit = 2 if paired else 1                      # reads per sample item
    if os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory() as td:
            def make(n, seed):
                fq = os.path.join(td, "s%d" % seed)
                if paired:
                    b = synth_pairs(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=seed)
                    write_fastq(b[0], fq + ".1.fq"); write_fastq(b[1], fq + ".2.fq")
                    return list(b), ["-1", fq + ".1.fq", "-2", fq + ".2.fq"]
                b = synth_reads(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=seed)
                write_fastq(b, fq + ".fq")
                return [b], [fq + ".fq"]

            def run(p, inputs, out=os.devnull, extra=()):
                t0 = time.perf_counter()
                r = subprocess.run([ref_bin, "--wrapper", "basic-0", "-t", "-p", str(p)] + args + list(extra) +
                                   ["-x", base] + inputs + [out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError("reference failed: " + r.stderr.decode(errors="replace")[-300:])
                m = re.search(r"Time searching: (\d+):(\d+):(\d+)", r.stderr.decode(errors="replace"))
                ts = int(m.group(1)) * 3600 + int(m.group(2)) * 60 + int(m.group(3)) if m else None
                return wall, ts

            nA = 200_000      # reads, or pairs: the sample whose SAM is diffed against the reference's (round 3: 40 000 pairs)
            bA, inA = make(nA, 4321)
            if diff_only:
                # the parity leg alone: the reference's SAM for the sample against the product's
                p = min(64, cores)
                sam_path = os.path.join(td, "ref.sam")
                w, _ = run(p, inA, out=sam_path, extra=["-S", "--sam-nohead", "--reorder"])
                with open(sam_path, "rb") as f:
                    want = f.read().split(b"\n")
                got = _gpu_sam(idx, A.make_policy(**pol), bA).split(b"\n")
                bad = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
                return {"value": nA * unit / w, "unit": "reads/s", "cores": p, "kind": "reference", "host_cores": cores,
                        "cpu_model": _cpu_model(), "reads_diffed_vs_reference": nA * unit, "diff_mismatches": bad,
                        "sample": "unmodified bowtie-align-s %s -p %d on %d reads, index load included (diff only, no sweep)" % (" ".join(args), p, nA * unit)}
            # size the larger sample from a first run at all cores
            wA, _ = run(cores, inA)
            nB = int(min(2_000_000 // unit if paired else 4_000_000, max(3 * nA, seconds * nA / max(wA, 0.5))))
            if paired:
                nB = min(nB, 500_000)              # the numpy pair generator is a python loop
            bB, inB = make(nB, 4322)
            sweep = {}
            for p in sorted({min(32, cores), min(64, cores), min(128, cores), cores}):
                w1, _ = run(p, inA)
                w2, ts2 = run(p, inB)
                rate = (nB - nA) * unit / (w2 - w1) if w2 > 1.15 * w1 else nB * unit / w2
                sweep[p] = {"reads_per_s": rate, "n1": nA * unit, "t1": w1, "n2": nB * unit, "t2": w2, "time_searching_s": ts2}
            best_p = max(sweep, key=lambda k: sweep[k]["reads_per_s"])
            # one core
            n1 = max(2000, int(nA * 0.02))
            b1, in1 = make(n1, 4323)
            w1a, _ = run(1, in1)
            n1b = 3 * n1
            b1b, in1b = make(n1b, 4324)
            w1b, _ = run(1, in1b)
            p1 = (n1b - n1) * unit / (w1b - w1a) if w1b > 1.15 * w1a else n1b * unit / w1b
            out = {"value": sweep[best_p]["reads_per_s"], "unit": "reads/s", "cores": best_p, "kind": "reference",
                   "host_cores": cores, "cpu_model": _cpu_model(), "p1_value": p1,
                   "time_searching_s": sweep[best_p]["time_searching_s"], "p_sweep": sweep,
                   "sample": "unmodified bowtie-align-s %s on the same index, -p sweep; per setting %d and %d reads, "
                             "value = (n2-n1)/(t2-t1) of the fastest setting (-p %d)" % (" ".join(args), nA * unit, nB * unit, best_p)}
            # parity against the reference itself on the first sample (output order: --reorder = input order)
            if idx is not None:
                sam_path = os.path.join(td, "ref.sam")
                run(best_p, inA, out=sam_path, extra=["-S", "--sam-nohead", "--reorder"])
                with open(sam_path, "rb") as f:
                    want = f.read().split(b"\n")
                got = _gpu_sam(idx, A.make_policy(**pol), bA).split(b"\n")
                bad = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
                out["reads_diffed_vs_reference"] = nA * unit
                out["diff_mismatches"] = bad
                if bad:
                    for a, b in zip(got, want):
                        if a != b:
                            out["first_diff"] = {"got": a.decode(errors="replace")[:300], "want": b.decode(errors="replace")[:300]}
                            break
            return out
    # port: the C restatement, one core
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as OL
    import refrun as R
    oi = OL.OracleIndex(base)
    opol = OL.make_policy(**pol)
    n = 2000
    t0 = time.perf_counter()
    if paired:
        b1, b2 = synth_pairs(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=4321)
        R.oracle_search_pairs(oi, opol, b1, b2)
    else:
        batch = synth_reads(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=4321)
        R.oracle_search(oi, opol, batch)
    dt = time.perf_counter() - t0
    return {"value": n * unit / dt, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": _cpu_model(),
            "sample": "%d synthetic %d-bp reads through oracle/bt_oracle.c via ctypes, %.2fs" % (n * unit, wl["length"], dt)}


def launch_ranks(args) -> int:
    """Re-run this command line as `--gpus` ranks of one node under torch.distributed.run (one process per GPU, rendezvous
    on 127.0.0.1, a free port); rank 0's JSON line is this process's output.  Returns the launcher's exit status."""
    import socket
    import subprocess
    if not args.dry_ranks and args.dist_backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print("bench.py: --gpus %d over RCCL needs %d GPUs, this node shows %d" % (args.gpus, args.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL on this host: dmabuf IPC only
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def reduce_counters(dist, cnt5, wall_t, world, backend):
    """The run's only collective (hit.h:169-175, 280-289: HitSink's counters are per process): SUM of the per-rank hit
    counters and MAX of the per-rank wall time -- RCCL over xGMI with the nccl backend -- checked against the per-rank
    counters gathered one by one.  Returns (summed counters, max wall)."""
    if backend != "nccl":
        cnt5, wall_t = cnt5.cpu(), wall_t.cpu()
    dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    mine = cnt5.clone()
    dist.all_reduce(cnt5, op=dist.ReduceOp.SUM)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if not torch.equal(torch.stack(parts).sum(dim=0), cnt5):
        raise SystemExit("bench.py: the all-reduced hit counters differ from the sum of the per-rank counters")
    return cnt5, float(wall_t[0].item())


def dry_ranks(args, rank, world):
    """--dry-ranks: everything a rank does around the search -- rendezvous, the counter reduce and its check, rank 0's
    line -- with made-up counters and no GPU."""
    import torch.distributed as dist
    backend = "gloo" if not torch.cuda.is_available() else args.dist_backend
    if world > 1:
        dist.init_process_group(backend)
    cnt5 = torch.tensor([100 + rank, 90 + rank, 0, 10 + rank, rank, 110 + 2 * rank, 0], dtype=torch.int64)
    wall_t = torch.tensor([1.0 + 0.25 * rank], dtype=torch.float64)
    wall = float(wall_t[0])
    if world > 1:
        if backend == "nccl":
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            cnt5, wall_t = cnt5.to(dev), wall_t.to(dev)
        cnt5, wall = reduce_counters(dist, cnt5, wall_t, world, backend)
    if rank == 0:
        c5 = [int(x) for x in cnt5.tolist()]
        print(json.dumps({"metric": "aligned reads/sec (whole node)", "dry": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "value": (c5[0] + c5[4]) * args.steps / wall, "reads_processed_per_s": c5[5] * args.steps / wall,
                          "unit": "reads/s", "scaling": args.scaling, "dist_backend": backend,
                          "config": {"hit_counters_last_step": {"aligned": c5[0], "reported": c5[1], "reported_paired": c5[2],
                                                                "unaligned": c5[3], "maxed": c5[4]}, "reads": c5[5]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("BT_WORKLOAD", "big_n2_100"))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (0 = workload default)")
    ap.add_argument("--genome", type=int, default=int(os.environ.get("BT_GENOME_BP", "0")),
                    help="synthetic genome length for the big_* workloads (0 = hg19 scale)")
    ap.add_argument("--pipes", type=int, default=1, help="contexts/streams the steps are pipelined over")
    ap.add_argument("--carry", type=int, default=-1,
                    help="bt_ctx_set_carry: launches a read may ride along with the steps after its own.  Default (-1): 12 "
                         "for steps of fewer than 64 M reads per GPU (it doubles the rate of 16 M-read steps, "
                         "profiles/README.md), 0 above that (every step runs to its last read; carry-over gains nothing at "
                         "200 M reads per step).")
    ap.add_argument("--no-carry", action="store_true", help="same as --carry 0")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", dest="verify", action="store_false",
                    help="skip re-checking every reported hit of the last step against the genome (bowtie_amd/verify.py; unpaired workloads)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the workload's reads per GPU (default); strong: that many reads in total, sharded over the GPUs")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--iters-hist", action="store_true", help="print the per-read LF-round distribution (diagnostics)")
    ap.add_argument("--env-sweep", default="",
                    help="diagnostics: after the measurement, the same workload again (same process, index and reads resident) under each "
                         "of these environments -- 'label:NAME=V,NAME=V;label2:...' -- e.g. the gates of bt_best_kernel; one rank only")
    ap.add_argument("--also", default="auto",
                    help="comma-separated workloads to run afterwards (2 steps each, own process, SAM diff against the reference) "
                         "and report under config.other_workloads; 'auto' = BASELINE configs 3 and 5 (big_v2_76, "
                         "big_pe_n1_best_50) after the default single-GPU run, nothing otherwise; 'none' = nothing")
    ap.add_argument("--cpu-diff-only", action="store_true",
                    help="CPU leg: only the SAM diff of a sample against the reference binary (no -p sweep)")
    ap.add_argument("--no-strong", action="store_true",
                    help="with --gpus N > 1 and weak scaling: skip the second, strong-scaling measurement (config.strong)")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="no GPU work: start the ranks, reduce made-up per-rank hit counters the way the real run does, print the "
                         "line's skeleton (CPU test of the launch + reduce path)")
    ap.add_argument("--kernel-sha", action="store_true", help="print the source digests profiles/traffic.json entries are keyed by, and exit")
    args = ap.parse_args()
    if args.kernel_sha:
        print(json.dumps({k: kernel_source_sha16(k) for k in KERNEL_SOURCES}))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own (the driver's command shape): start one rank per GPU ourselves, the way
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...` would
        raise SystemExit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_ranks:
        return dry_ranks(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if world > 1 and args.dist_backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, this node shows %d" % (world, world, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()      # several ranks may share a GPU (gloo, tests)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    wl = WORKLOADS[args.workload]
    n = args.reads or wl["reads"]
    n_total = n
    shard = None
    if args.scaling == "strong":
        # BASELINE config 4 as written: the reads are one pool, sharded contiguously by rank
        per = n // world
        lo = rank * per + min(rank, n % world)
        n = per + (1 if rank < n % world else 0)
        shard = (lo, lo + n)
    L = wl["length"]

    # ---- index: in HBM before anything is timed ------------------------------------------------
    t0 = time.perf_counter()
    if wl["index"] == "ecoli":
        base = os.path.join(ROOT, "tests", "golden", "e_coli")
        text_np = AL.restore_text(base)
        index_note = "e_coli (bundled reference index, 4.9 Mbp)"
    else:
        from bowtie_amd import ebwt_build as EB
        base, text_np, index_note = EB.ensure_big_index(args.genome, dev, rank, world)
    idx = AL.Index(base, need_mirror=True, device=local)
    text_t = torch.from_numpy(text_np).to(dev)
    log("[bench] index %s loaded in %.1fs" % (index_note, time.perf_counter() - t0))

    # ---- reads: synthetic, generated straight into HBM, sharded by rank -------------------------
    t0 = time.perf_counter()
    paired = bool(wl.get("paired"))
    rb2 = None
    if paired:
        rb, rb2 = synth_pairs_torch(text_t, n, L, mm_dist=wl["mm_dist"], seed=1000 + rank, first_id=rank * n)
    else:
        if shard is not None:
            rb = synth_reads_torch(text_t, n_total, L, mm_dist=wl["mm_dist"], seed=1000, shard=shard)
        else:
            rb = synth_reads_torch(text_t, n, L, mm_dist=wl["mm_dist"], seed=1000 + rank, first_id=rank * n)
    if not args.verify:
        del text_t
    hit_cap = 2 if paired else 1
    pol = A.make_policy(**wl["pol"])
    lib = AL.lib()
    # Software pipelining across steps: `--pipes` contexts, each with its own HIP stream, scratch and
    # output buffers, take the steps round-robin.  A step's last few long-running reads (the
    # backtracking tail) then drain while the next step's wavefronts already fill the machine --
    # what a host driver double-buffering read batches does.
    # Carry-over (bt_ctx_set_carry, include/bowtie_amd.h): the reads still being searched when a step's reads have
    # all been handed out are parked and resumed by the context's next step, so a step's results are complete when
    # the next step (or the closing bt_ctx_sync, inside the timed region) is.  Each context therefore alternates
    # between two sets of output arrays.
    def carry_for(n_use):
        # default: where the backtracking tail is long (-n modes) and the step is small enough for it to matter; -v steps
        # finish within a few hundred rounds of each other and only pay for the closing launch (big_v2_76, 50 M reads:
        # 28.4 M reads/s with carry-over against 36 M without, profiles/r3)
        age = 0 if args.no_carry else (args.carry if args.carry >= 0 else (12 if (n_use < 64_000_000 and wl["pol"]["mode"] == "n") else 0))
        if paired or wl["pol"].get("best") or L > 112:
            age = 0
        return min(age, 12)

    if paired and lib.bt_index_load_reference(idx._h) != 0:
        raise RuntimeError("bt_index_load_reference failed")
    iters_t = torch.zeros(n, dtype=torch.int32, device=dev) if args.iters_hist else None

    def measure(n_use, steps, warmup):
        """`warmup` untimed + `steps` timed steps over the first n_use reads of this rank's buffer: contexts, output
        arrays, the barrier + synchronize bracket, op counters of the timed steps."""
        carry_age = carry_for(n_use)
        carry = carry_age > 0
        mm_cap = n_use * 8
        pipes = []
        for pi in range(max(1, args.pipes)):
            st = torch.cuda.current_stream() if pi == 0 else torch.cuda.Stream()
            o = dict(stream=st, busy=False, sets=[])
            # a step's outputs stay in use until the last of its reads is done: carry_age launches later at most
            for si in range(min(carry_age + 1, max(steps, warmup, 1)) if carry else 1):
                d = dict(hits=torch.zeros(n_use * hit_cap * 24, dtype=torch.uint8, device=dev),
                         n_hits=torch.zeros(n_use, dtype=torch.int32, device=dev),
                         status=torch.zeros(n_use, dtype=torch.uint8, device=dev),
                         mm_pool=torch.zeros(mm_cap, dtype=torch.int16, device=dev))
                d["hbc"] = A.HitBatchC(hit_cap, d["hits"].data_ptr(), d["n_hits"].data_ptr(), d["status"].data_ptr(),
                                       d["mm_pool"].data_ptr(), mm_cap, 0)
                o["sets"].append(d)
            o["al"] = AL.Aligner(idx, pol, stream=st.cuda_stream)
            if os.environ.get("BT_LOCUS_OFF") == "1":        # A/B: this measurement's launches stay in row space (--env-sweep "rowspace:BT_LOCUS_OFF=1")
                lib.bt_ctx_set_locus(o["al"]._h, 0)
            if carry and lib.bt_ctx_set_carry(o["al"]._h, carry_age) != 0:
                raise RuntimeError("bt_ctx_set_carry failed")
            lib.bt_ctx_set_max_read_len(o["al"]._h, L)         # synthetic reads: all of length L (rows are padded to 16)
            pipes.append(o)
        torch.cuda.synchronize()
        rbc = A.ReadBatchC(n_use, rb["stride"], rb["seq"].data_ptr(), rb["qual"].data_ptr(), rb["len"].data_ptr(),
                           rb["seed"].data_ptr())
        rbc2 = None
        if paired:
            rbc2 = A.ReadBatchC(n_use, rb2["stride"], rb2["seq"].data_ptr(), rb2["qual"].data_ptr(), rb2["len"].data_ptr(),
                                rb2["seed"].data_ptr())
        if iters_t is not None:
            lib.bt_ctx_set_iters_buffer(pipes[0]["al"]._h, iters_t.data_ptr())
        kernel_ms, flush_ms = [], []
        last = {}

        def retire(o):
            if o["busy"]:
                if lib.bt_ctx_sync(o["al"]._h) != 0:      # with carry-over: finishes the parked reads first
                    raise RuntimeError("bt_ctx_sync failed")
                nl = C.c_uint32()
                lib.bt_ctx_span_ms(o["al"]._h, C.byref(nl))
                for i in range(max(0, nl.value - 16), nl.value):       # HIP events on the kernel's stream, per launch
                    kernel_ms.append(float(lib.bt_ctx_launch_ms(o["al"]._h, i)))
                flush_ms.append(float(lib.bt_ctx_launch_ms(o["al"]._h, -1)))
                o["busy"] = False

        def step(k):
            o = pipes[k % len(pipes)]
            if not carry:
                retire(o)
            d = o["sets"][(k // len(pipes)) % len(o["sets"])]
            if paired:
                rc = lib.bt_align_pairs_device(o["al"]._h, C.byref(rbc), C.byref(rbc2), C.byref(d["hbc"]), None)
            else:
                rc = lib.bt_align_batch_device(o["al"]._h, C.byref(rbc), C.byref(d["hbc"]), None)
            if rc != 0:
                raise RuntimeError("bt_align_batch_device: " + AL.strerror(rc))
            o["busy"] = True
            last["set"] = d

        def barrier():
            for o in pipes:
                retire(o)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for k in range(warmup):
            step(k)
        barrier()
        cnt = A.OpCounts()
        for o in pipes:
            lib.bt_ctx_counts(o["al"]._h, C.byref(cnt), 1)        # reset: count the timed steps only
        kernel_ms.clear()
        flush_ms.clear()
        prof = (C.c_ulonglong * 96)()
        has_prof = hasattr(lib, "bt_best_prof_read") and lib.bt_best_prof_read(prof, 96, 1) > 0      # reset: the timed steps only
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        barrier()
        wall = time.perf_counter() - t0
        if has_prof:
            # the profiling build of bt_best_kernel (make bestprof): wavefront cycles per section of the engine
            nsec = lib.bt_best_prof_read(prof, 96, 0)
            names = ["RUN", "BEGIN", "SETQ", "ADV", "LEAF", "STREAK", "CURTAIL", "SPLIT", "SORT", "CHASE", "REPORT", "REF", "END", "FRONT",
                     "HOT", "HOT_STEP", "HOT_SEND", "HOT_CHASE", "COLD", "COLD_TAKE", "COLD_EXIT", "COLD_POST", "COLD_RUN", "COLD_PRE"]
            tot = max(1, prof[0])
            SECTIONS.clear()
            for i in range(min(nsec, len(names))):
                cyc, passes, lanes = prof[3 * i], prof[3 * i + 1], prof[3 * i + 2]
                SECTIONS[names[i]] = {"cycles": cyc, "share_of_run": cyc / tot, "passes": passes, "lanes_per_pass": lanes / max(1, passes)}
                log("[bench] section %-8s %6.1f%% of RUN  %12d passes  %5.1f lanes/pass  %8.0f cycles/pass" %
                    (names[i], 100.0 * cyc / tot, passes, lanes / max(1, passes), cyc / max(1, passes)))
        c = {}
        for o in pipes:
            lib.bt_ctx_counts(o["al"]._h, C.byref(cnt), 0)
            for kk, v in cnt.as_dict().items():
                c[kk] = c.get(kk, 0) + v
            if hasattr(lib, "bt_ctx_jump_counts"):
                lk, stp = C.c_uint64(), C.c_uint64()
                lib.bt_ctx_jump_counts(o["al"]._h, C.byref(lk), C.byref(stp))
                c["jump_lookups"] = c.get("jump_lookups", 0) + lk.value
                c["jump_steps"] = c.get("jump_steps", 0) + stp.value
        return dict(wall=wall, c=c, kernel_ms=kernel_ms, flush_ms=flush_ms, last=last, pipes=pipes, carry_age=carry_age)

    log("[bench] %d x %d-bp reads in HBM in %.1fs" % (n, L, time.perf_counter() - t0))
    M = measure(n, args.steps, args.warmup)
    wall, c, kernel_ms, flush_ms, last, pipes, carry_age = M["wall"], M["c"], M["kernel_ms"], M["flush_ms"], M["last"], M["pipes"], M["carry_age"]
    log("[bench] main measurement: %.3f M reads/s on this rank, %.1f ms per step, kernel %.1f ms" %
        (n * (2 if paired else 1) * args.steps / wall / 1e6, 1e3 * wall / args.steps, sum(kernel_ms) / max(1, len(kernel_ms))))

    verified = None
    if args.verify:
        # size-independent parity property at full size: every hit is re-derived from the text
        from bowtie_amd import verify as V
        tl, plen, rstarts = V.read_fragments(base)
        o = last["set"]
        t1 = time.perf_counter()
        if paired:
            # pairs: both mates' windows and mismatch lists + same reference, upstream mate first, orientation, fragment
            # length, containment (bowtie_amd/verify.py: verify_pairs); `checked` counts pairs
            verified = V.verify_pairs(text_t, tl, rstarts, rb["seq"], rb["qual"], rb2["seq"], rb2["qual"], L, L, o["hits"],
                                      o["n_hits"], o["mm_pool"], dict(wl["pol"], seed_len=28, qual_thresh=70), hit_cap=hit_cap)
        else:
            verified = V.verify_hits(text_t, tl, rstarts, rb["seq"], rb["qual"], L, o["hits"], o["n_hits"], o["mm_pool"],
                                     dict(wl["pol"], seed_len=28, qual_thresh=70))
        log("[bench] verify: %s in %.1fs" % (verified, time.perf_counter() - t1))
        if any(v for k, v in verified.items() if k != "checked"):
            raise SystemExit("bench.py --verify: reported hits fail the re-check: %s" % verified)
    if iters_t is not None and rank == 0:
        it = iters_t.to(torch.float64)
        qs = torch.quantile(it[:min(n, 4_000_000)], torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999, 1.0], dtype=torch.float64, device=dev))
        log("[bench] LF rounds per read: mean %.1f  p50/p90/p99/p99.9/p99.99/max = %s  (reads > 20k rounds: %d, their share of all rounds %.1f%%)" %
            (it.mean().item(), [int(x) for x in qs.tolist()], int((it > 20000).sum().item()),
             100.0 * it[it > 20000].sum().item() / max(1.0, it.sum().item())))
    n_hits, status = last["set"]["n_hits"], last["set"]["status"]
    aligned = int((n_hits > 0).sum().item())
    bad = int(((status & (A.BT_ST_OVERFLOW | A.BT_ST_MMPOOL)) != 0).sum().item())
    # HitSink's five counters (hit.h:169-175, 280-289) for this rank's shard of the last step, + reads and
    # reads the product could not finish: one int64 all-reduce over xGMI is the whole collective
    mult = 2 if paired else 1
    mh = int(pol.mhits)
    maxv = 0xFFFFFFFF if mh == 0xFFFFFFFF else mh * mult
    nh = n_hits.to(torch.int64) & 0xFFFFFFFF
    is_max = nh > maxv
    is_al = (nh > 0) & ~is_max
    kk = (0x7FFFFFFF if pol.all_hits else int(pol.khits) * mult)
    rep = torch.where(is_al, nh.clamp(max=kk), torch.zeros_like(nh)).sum()
    cnt5 = torch.stack([is_al.sum(), rep if not paired else torch.zeros_like(rep), rep if paired else torch.zeros_like(rep),
                        (nh == 0).sum(), is_max.sum(), torch.tensor(n, device=dev), torch.tensor(bad, device=dev)]).to(torch.int64)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        cnt5, wall = reduce_counters(dist, cnt5, wall_t, world, args.dist_backend)      # the hit-count reduce
    c5 = [int(x) for x in cnt5.tolist()]
    aligned_all, reads_all, bad_all = float(c5[0] + c5[4]), float(c5[5]), float(c5[6])

    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong:
        # the other regime in the same line: the workload's reads in total (BASELINE config 4 as written: 200 M reads
        # over the node), sharded -- every rank takes the first n_total / world reads of its own buffer
        n_s = max(1, n_total // world)
        for o in pipes:
            o["sets"].clear()
        last.clear()
        torch.cuda.empty_cache()
        Ms = measure(n_s, args.steps, args.warmup)
        ws = torch.tensor([Ms["wall"]], dtype=torch.float64, device=dev)
        if args.dist_backend != "nccl":
            ws = ws.cpu()
        dist.all_reduce(ws, op=dist.ReduceOp.MAX)
        wall_s = float(ws[0].item())
        strong = {"reads_total_per_step": n_s * world * mult, "reads_per_gpu_per_step": n_s * mult,
                  "reads_processed_per_s": n_s * world * mult * args.steps / wall_s,
                  "value": n_s * world * mult * args.steps / wall_s * (aligned_all / max(1.0, reads_all)), "unit": "reads/s",
                  "value_counts": "aligned reads/s = reads processed/s x the weak run's aligned fraction (same read generator)",
                  "ms_per_step": wall_s * 1e3 / args.steps,
                  "carry_over_launches": Ms["carry_age"]}

    if rank == 0:
        per_launch = {k: v / max(1, args.steps) for k, v in c.items()}
        # a step's launch time: its own launches (main kernel + second pass) plus its share of the closing launch
        # that finishes the reads carried out of the last step
        kmain = sum(kernel_ms) / max(1, len(kernel_ms))
        kavg = kmain + sum(flush_ms) / max(1, args.steps)
        if kavg <= 0:                     # a library without per-launch events: the step's wall time stands in
            kavg = kmain = wall * 1e3 / max(1, args.steps)
        kname = (lib.bt_ctx_last_kernel_name(pipes[0]["al"]._h) or b"").decode()
        tr = measured_traffic(kname, args.workload)
        abytes = algorithmic_bytes(per_launch, n * (2 if paired else 1), L, aligned * (2 if paired else 1))
        achieved = abytes / (kavg * 1e-3) / 1e9
        out = {
            # `value` is the metric as worded: reads that aligned, per second, over all GPUs (a pair = 2 reads);
            # `reads_processed_per_s` counts every read put through the aligner, aligned or not (rounds 1-3 reported that as `value`)
            "metric": "aligned reads/sec (whole node)", "value": aligned_all * mult * args.steps / wall,
            "unit": "reads/s", "aligned_reads_per_s": aligned_all * mult * args.steps / wall,
            "reads_processed_per_s": reads_all * mult * args.steps / wall,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": args.workload, "index": index_note, "read_len": L,
                       "policy": wl["pol"], "reads_per_gpu_per_step": n * (2 if paired else 1),
                       "value_counts": "reads with at least one reported alignment, or more than -m allows (a pair = 2 reads); reads_processed_per_s counts every read",

                       "pairs_per_gpu_per_step": n if paired else None,
                       "reads_with_alignment_per_s": aligned_all * mult * args.steps / wall,
                       "pct_aligned": 100.0 * aligned_all / reads_all, "reads_overflowed": bad_all,
                       "hit_counters_last_step": {"aligned": c5[0], "reported": c5[1], "reported_paired": c5[2],
                                                  "unaligned": c5[3], "maxed": c5[4]},
                       "pipelined_contexts": len(pipes),
                       "hits_verified_against_text": verified["checked"] if verified else None,
                       "verified_unit": ("pairs (both mates + pair constraints)" if paired else "hits") if verified else None,
                       "parallelism": "reads sharded x%d, index replicated" % world,
                       "strong": strong},
            "best_sections": SECTIONS or None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # PMC traffic, per launch: with the guide's gfx950 correction (FETCH_SIZE x 2) -- an upper bound for
                         # 32-byte gathers -- and as the counters tally it (x 1, what the calibration on known-count gathers says)
                         "traffic": (tr["hbm_bytes_per_read"] * n * mult if tr and not args.genome else None),
                         "traffic_as_tallied": (tr["hbm_bytes_per_read_as_tallied"] * n * mult if tr and not args.genome and tr.get("hbm_bytes_per_read_as_tallied") else None),
                         "traffic_note": (tr["source"] if tr else "no rocprofv3 PMC profile of this round's source of this kernel on this workload in profiles/traffic.json (round 3 measured 213 KB/read on big_n2_100, an upper bound: 1.9 x algorithmic)"),
                         "achieved_over_wall": abytes * args.steps / wall / 1e9,
                         "kernel": kname, "kernel_ms_avg": kavg, "kernel_ms_main_avg": kmain,
                         # a step of the stateful path is two dispatches of the same kernel: the main launch and the second pass over
                         # the reads that outgrew their arena (it returns at once when there are none) -- rocprofv3's per-kernel
                         # average is over both, kernel_ms_avg is the step's span (HIP events around both)
                         "dispatches_per_step": 2 if (paired or wl["pol"].get("best")) else 1,
                         "carry_over_launches": carry_age, "flush_ms_total": sum(flush_ms),
                         "reads_searched_again_last_step": sum(int(lib.bt_ctx_last_retried(o["al"]._h)) for o in pipes),
                         "algorithmic_bytes_per_launch": abytes,
                         "ops_per_read": {k: per_launch[k] / n for k in ("lfex", "lf2", "lf1", "chase", "frames", "rescans", "cand_scans", "fetches",
                                                                          "loc_lfex", "loc_lf1", "loc_chase", "loc_records", "loc_windows")},
                         "locus_mode": bool(lib.bt_ctx_get_locus(pipes[0]["al"]._h)),
                         # the jump table (csrc/bt_rank.h): look-ups per read among the searches, and the LF steps of the reference's
                         # algorithm that lay behind them (counted in ops_per_read's lf2 / lf1 all the same)
                         "jump_table_GB": (lib.bt_index_jump_bytes(idx._h) / 1e9) if hasattr(lib, "bt_index_jump_bytes") else 0.0,
                         "jump_lookups_per_read": per_launch.get("jump_lookups", 0) / n, "jump_steps_per_read": per_launch.get("jump_steps", 0) / n,
                         "locus_image_GB": lib.bt_index_locus_bytes(idx._h) / 1e9, "locus_image_build_s": lib.bt_index_locus_build_seconds(idx._h),
                         "lane_iters_per_read": per_launch["lane_iters"] / n,
                         "mean_active_lanes_per_round": per_launch["lane_iters"] / max(1.0, per_launch["wave_rounds"]),
                         "wave_rounds_per_launch": per_launch["wave_rounds"]},
        }
        gc = None
        if not paired and not wl["pol"].get("best"):
            try:
                gal = AL.Aligner(idx, pol)
                gc = gather_ceiling(wl["index"], gal, torch.cuda.get_device_properties(dev).multi_processor_count)
                gal.close()
            except Exception:       # noqa: BLE001
                gc = None
        if gc is None and not args.genome:
            gc = gather_ceiling(wl["index"])
        if gc:
            # the kernel's rank traffic (the 128-byte gathers) against what the memory system delivers for that
            # pattern alone; `frac` above stays against the 8 TB/s streaming peak
            out["roofline"]["gather_ceiling_GBps"] = gc["GBps"]
            out["roofline"]["frac_of_gather_ceiling"] = achieved / gc["GBps"]
            out["roofline"]["gather_ceiling_source"] = gc["source"]
        if not args.no_cpu:
            cb = cpu_baseline(base, wl, text_np, idx, diff_only=args.cpu_diff_only)
            cb["value_counts"] = "reads processed per second; the reference's results are the product's (diff_mismatches), so the same fraction aligns"
            cb["aligned_reads_per_s"] = cb["value"] * aligned_all / max(1.0, reads_all)
            out["cpu_baseline"] = cb
            # (`vs_baseline` stays null: BASELINE.md holds no published number for this metric.  This is the run's own ratio:
            # reads processed per second here over the reference's at its best -p on this host's cores, same reads, same results)
            if cb.get("value"):
                out["vs_cpu_baseline"] = out["reads_processed_per_s"] / cb["value"]
            out["config"]["reads_diffed_vs_reference"] = cb.get("reads_diffed_vs_reference")
            out["config"]["diff_mismatches"] = cb.get("diff_mismatches")
        if args.env_sweep and world == 1:
            # A/B of library settings that are read at launch time (BT_BEST_*): contexts of the main measurement closed first,
            # so that every setting's contexts find the same free memory
            ref_hits = int(last["set"]["n_hits"].to(torch.int64).sum().item())
            for o in pipes:
                o["al"].close()
            out["env_sweep"] = []
            for item in [x for x in args.env_sweep.split(";") if x]:
                label, _, kvs = item.partition(":")
                kv = dict(x.split("=", 1) for x in kvs.split(",") if x)
                old_env = {k: os.environ.get(k) for k in kv}
                os.environ.update(kv)
                try:
                    Ms = measure(n, max(1, min(args.steps, 2)), 1)
                finally:
                    for k, v in old_env.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                hs = int(Ms["last"]["set"]["n_hits"].to(torch.int64).sum().item())
                rec = {"label": label, "env": kv, "reads_processed_per_s": n * mult * max(1, min(args.steps, 2)) / Ms["wall"],
                       "kernel_ms_avg": sum(Ms["kernel_ms"]) / max(1, len(Ms["kernel_ms"])), "n_hits_sum_equal": hs == ref_hits}
                out["env_sweep"].append(rec)
                log("[bench] env-sweep %-22s %8.3f M reads/s  kernel %9.1f ms  same hit count: %s" %
                    (label, rec["reads_processed_per_s"] / 1e6, rec["kernel_ms_avg"], rec["n_hits_sum_equal"]))
                for o in Ms["pipes"]:
                    o["al"].close()
                del Ms
                torch.cuda.empty_cache()
        also = args.also
        if also == "auto":
            # every other BASELINE configuration (2: e_coli -v 0 36 bp; 3: -v 2 76 bp; 5's share: pairs --best) and the two other
            # stateful modes people run: single-end --best, and pairs without --best (the reference's default paired aligner)
            also = ("big_v2_76,big_pe_n1_best_50,ecoli_v0_36,big_n2_best_100:16000000,big_pe_n1_50_v1:6250000"
                    if (world == 1 and args.workload == "big_n2_100" and not args.reads and not args.genome and not args.no_cpu) else "none")
        if also != "none" and world == 1:
            # BASELINE configs 3 and 5 in the same line: each in its own process (the index cache under /tmp is reused),
            # 2 timed steps, every hit re-verified where the workload allows it, a sample diffed against the reference
            import subprocess
            # this process's contexts and index replica go first: the locus image alone is 104 GB at hg19 scale, and a child
            # that finds the device that full searches without one (round 5's final call: config 5's share 10.1 M reads/s
            # in this leg against 11.7 M on its own)
            for o in pipes:
                o["al"].close()
            idx.close()
            del rb, rb2, M, pipes, last
            torch.cuda.empty_cache()
            out["config"]["other_workloads"] = {}
            for item in [x for x in also.split(",") if x]:
                name, _, nreads = item.partition(":")           # "workload[:reads per step]"
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "2", "--warmup", "1",
                                    "--cpu-diff-only", "--also", "none"] + (["--reads", nreads] if nreads else []),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                try:
                    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
                    out["config"]["other_workloads"][name] = {
                        "value": d["value"], "unit": d["unit"], "aligned_reads_per_s": d.get("aligned_reads_per_s"),
                        "ms_per_step": d["ms_per_step"], "steps": d["steps"], "reads_per_gpu_per_step": d["config"]["reads_per_gpu_per_step"],
                        "reads_processed_per_s": d.get("reads_processed_per_s"),
                        "roofline_frac": d["roofline"]["frac"], "kernel": d["roofline"]["kernel"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                        "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"],
                        "traffic": d["roofline"].get("traffic"), "traffic_as_tallied": d["roofline"].get("traffic_as_tallied"),
                        "hits_verified_against_text": d["config"].get("hits_verified_against_text"),
                        "reads_diffed_vs_reference": d["config"].get("reads_diffed_vs_reference"),
                        "diff_mismatches": d["config"].get("diff_mismatches"),
                        "cpu_reference_reads_per_s_incl_index_load": d.get("cpu_baseline", {}).get("value")}
                except (IndexError, ValueError, KeyError) as e:
                    out["config"]["other_workloads"][name] = {"error": "%s: %s" % (type(e).__name__, r.stderr.decode(errors="replace")[-300:])}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
