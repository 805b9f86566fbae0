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
# HBM-side bytes per read of bt_search_kernel, measured with rocprofv3 PMC (separate --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes over a 16 M-read launch of the same workload, profiles/r1_final/pmc_big_n2_100_16M.txt)
# and corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies this kernel's 128-byte
# side-pair fetches at 64 B (calibrated on the probe kernel, whose bytes are known:
# profiles/r1_final/calib_fetch_size.txt), so it is doubled; WRITE_SIZE calibrates exact.
#   (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / reads = (2 x 1.297e9 + 7.11e8) x 1024 / 16e6 = 212 KB/read
# bench.py cannot run the profiler on itself, so `traffic` is that per-read figure times the reads of one
# launch; null for workloads not profiled.
MEASURED_HBM_BYTES_PER_READ = {"big_n2_100": (2 * 1.297e9 + 7.11e8) * 1024.0 / 16_000_000}

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


def cpu_baseline(base: str, wl: dict, text_np: np.ndarray, seconds: float = 12.0):
    """Reference bowtie (oracle/_ref, unmodified, all host cores) on a bounded FASTQ sample of the
    same workload; falls back to the single-core C restatement if the binary is not on the box."""
    import subprocess
    import tempfile
    from bowtie_amd.synth import synth_reads, write_fastq
    pol = wl["pol"]
    args = ["-v", str(pol["mms"])] if pol["mode"] == "v" else ["-n", str(pol["mms"]), "-l", "28", "-e", "70"]
    if pol.get("best"):
        args.append("--best")
    paired = bool(wl.get("paired"))
    if paired:
        args += ["-X", str(pol.get("max_ins", 250))]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")
    cores = os.cpu_count() or 1
    if os.path.exists(ref_bin):
        # two runs of different size: the difference cancels the fixed cost (index load from the
        # page cache, spawning `cores` threads), leaving the reference's search throughput
        runs = []
        with tempfile.TemporaryDirectory() as td:
            n = 200_000
            for attempt in range(2):
                fq = os.path.join(td, "s.fq")
                if paired:
                    from bowtie_amd.synth import synth_pairs
                    n = min(n, 400_000)                     # the numpy pair generator is a python loop
                    b1, b2 = synth_pairs(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=4321 + attempt)
                    write_fastq(b1, fq + ".1"); write_fastq(b2, fq + ".2")
                    inputs = ["-1", fq + ".1", "-2", fq + ".2"]
                else:
                    batch = synth_reads(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=4321 + attempt)
                    write_fastq(batch, fq)
                    inputs = [fq]
                t0 = time.perf_counter()
                p = subprocess.run([ref_bin, "--wrapper", "basic-0", "-p", str(cores)] + args +
                                   ["-x", base] + inputs + [os.devnull], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                wall = time.perf_counter() - t0
                if p.returncode != 0:
                    break
                runs.append((n, wall))
                n = int(min(8_000_000, max(600_000, 2.0 * seconds * n / wall)))
        if len(runs) == 2:
            (n1, t1), (n2, t2) = runs
            rate = (n2 - n1) / (t2 - t1) if t2 > 1.2 * t1 else n2 / t2
            return {"value": rate, "unit": "reads/s", "cores": cores, "kind": "reference",
                    "sample": "unmodified bowtie-align-s -p %d %s on the same index: %d reads in %.2fs and %d reads "
                              "in %.2fs wall (incl. index load); value = (n2-n1)/(t2-t1)" %
                              (cores, " ".join(args), n1, t1, n2, t2)}
    # port: the C restatement, one core
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as OL
    import refrun as R
    oi = OL.OracleIndex(base)
    opol = OL.make_policy(**pol)
    n = 2000
    batch = synth_reads(text_np, n, wl["length"], mm_dist=wl["mm_dist"], seed=4321)
    t0 = time.perf_counter()
    R.oracle_search(oi, opol, batch)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic %d-bp reads through oracle/bt_oracle.c via ctypes, %.2fs" % (n, wl["length"], dt)}


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
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps, re-check every reported hit of the last step against the genome (bowtie_amd/verify.py)")
    ap.add_argument("--iters-hist", action="store_true", help="print the per-read LF-round distribution (diagnostics)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    wl = WORKLOADS[args.workload]
    n = args.reads or wl["reads"]
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
        rb = synth_reads_torch(text_t, n, L, mm_dist=wl["mm_dist"], seed=1000 + rank, first_id=rank * n)
    if not args.verify:
        del text_t
    hit_cap = 2 if paired else 1
    mm_cap = n * 8
    pol = A.make_policy(**wl["pol"])
    lib = AL.lib()
    # Software pipelining across steps: `--pipes` contexts, each with its own HIP stream, scratch and
    # output buffers, take the steps round-robin.  A step's last few long-running reads (the
    # backtracking tail) then drain while the next step's wavefronts already fill the machine --
    # what a host driver double-buffering read batches does.
    pipes = []
    for pi in range(max(1, args.pipes)):
        st = torch.cuda.current_stream() if pi == 0 else torch.cuda.Stream()
        o = dict(stream=st,
                 hits=torch.zeros(n * hit_cap * 24, dtype=torch.uint8, device=dev),
                 n_hits=torch.zeros(n, dtype=torch.int32, device=dev),
                 status=torch.zeros(n, dtype=torch.uint8, device=dev),
                 mm_pool=torch.zeros(mm_cap, dtype=torch.int16, device=dev), busy=False)
        o["al"] = AL.Aligner(idx, pol, stream=st.cuda_stream)
        o["hbc"] = A.HitBatchC(hit_cap, o["hits"].data_ptr(), o["n_hits"].data_ptr(), o["status"].data_ptr(),
                               o["mm_pool"].data_ptr(), mm_cap, 0)
        pipes.append(o)
    torch.cuda.synchronize()
    log("[bench] %d x %d-bp reads in HBM in %.1fs" % (n, L, time.perf_counter() - t0))
    rbc = A.ReadBatchC(n, rb["stride"], rb["seq"].data_ptr(), rb["qual"].data_ptr(), rb["len"].data_ptr(),
                       rb["seed"].data_ptr())
    rbc2 = None
    if paired:
        rbc2 = A.ReadBatchC(n, rb2["stride"], rb2["seq"].data_ptr(), rb2["qual"].data_ptr(), rb2["len"].data_ptr(),
                            rb2["seed"].data_ptr())
        if lib.bt_index_load_reference(idx._h) != 0:
            raise RuntimeError("bt_index_load_reference failed")
    n_hits, status = pipes[0]["n_hits"], pipes[0]["status"]
    iters_t = None
    if args.iters_hist:
        iters_t = torch.zeros(n, dtype=torch.int32, device=dev)
        lib.bt_ctx_set_iters_buffer(pipes[0]["al"]._h, iters_t.data_ptr())
    kernel_ms = []

    def retire(o):
        if o["busy"]:
            if lib.bt_ctx_sync(o["al"]._h) != 0:
                raise RuntimeError("bt_ctx_sync failed")
            kernel_ms.append(float(lib.bt_ctx_last_kernel_ms(o["al"]._h)))   # HIP events on the kernel's stream
            o["busy"] = False

    def step(k):
        o = pipes[k % len(pipes)]
        retire(o)
        if paired:
            rc = lib.bt_align_pairs_device(o["al"]._h, C.byref(rbc), C.byref(rbc2), C.byref(o["hbc"]), None)
        else:
            rc = lib.bt_align_batch_device(o["al"]._h, C.byref(rbc), C.byref(o["hbc"]), None)
        if rc != 0:
            raise RuntimeError("bt_align_batch_device: " + AL.strerror(rc))
        o["busy"] = True

    def barrier():
        for o in pipes:
            retire(o)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    barrier()
    cnt = A.OpCounts()
    for o in pipes:
        lib.bt_ctx_counts(o["al"]._h, C.byref(cnt), 1)        # reset: count the timed steps only
    kernel_ms.clear()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    wall = time.perf_counter() - t0
    c = {}
    for o in pipes:
        lib.bt_ctx_counts(o["al"]._h, C.byref(cnt), 0)
        for kk, v in cnt.as_dict().items():
            c[kk] = c.get(kk, 0) + v

    verified = None
    if args.verify:
        # size-independent parity property at full size: every hit is re-derived from the text
        from bowtie_amd import verify as V
        tl, plen, rstarts = V.read_fragments(base)
        o = pipes[(args.steps - 1) % len(pipes)] if args.steps > 0 else pipes[0]
        t1 = time.perf_counter()
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
    aligned = int((n_hits > 0).sum().item())
    bad = int(((status & (A.BT_ST_OVERFLOW | A.BT_ST_MMPOOL)) != 0).sum().item())
    tot = torch.tensor([wall, float(aligned), float(n), float(bad)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)     # the hit-count reduce over xGMI
        wall = float(mx[0].item())
    aligned_all, reads_all, bad_all = float(tot[1].item()), float(tot[2].item()), float(tot[3].item())

    if rank == 0:
        per_launch = {k: v / max(1, args.steps) for k, v in c.items()}
        kavg = sum(kernel_ms) / len(kernel_ms)
        abytes = algorithmic_bytes(per_launch, n * (2 if paired else 1), L, aligned * (2 if paired else 1))
        achieved = abytes / (kavg * 1e-3) / 1e9
        out = {
            "metric": "aligned reads/sec (whole node)", "value": reads_all * (2 if paired else 1) * args.steps / wall,
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": args.workload, "index": index_note, "read_len": L,
                       "policy": wl["pol"], "reads_per_gpu_per_step": n * (2 if paired else 1),
                       "value_counts": "reads processed, aligned or not (a pair = 2 reads)",
                       "pairs_per_gpu_per_step": n if paired else None,
                       "reads_with_alignment_per_s": aligned_all * args.steps / wall,
                       "pct_aligned": 100.0 * aligned_all / reads_all, "reads_overflowed": bad_all,
                       "pipelined_contexts": len(pipes),
                       "hits_verified_against_text": verified["checked"] if verified else None,
                       "parallelism": "reads sharded x%d, index replicated" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (MEASURED_HBM_BYTES_PER_READ[args.workload] * n
                                     if args.workload in MEASURED_HBM_BYTES_PER_READ and not args.genome else None),
                         "traffic_note": "rocprofv3 (2 x FETCH_SIZE + WRITE_SIZE) per read, gfx950-calibrated (profiles/r1_final/calib_fetch_size.txt), x reads per launch",
                         "achieved_over_wall": abytes * args.steps / wall / 1e9,
                         "kernel": "bt_best_kernel" if wl["pol"].get("best") else "bt_search_kernel", "kernel_ms_avg": kavg,
                         "algorithmic_bytes_per_launch": abytes,
                         "ops_per_read": {k: per_launch[k] / n for k in ("lfex", "lf2", "lf1", "chase", "frames", "rescans", "cand_scans", "fetches")},
                         "lane_iters_per_read": per_launch["lane_iters"] / n,
                         "mean_active_lanes_per_round": per_launch["lane_iters"] / max(1.0, per_launch["wave_rounds"]),
                         "wave_rounds_per_launch": per_launch["wave_rounds"]},
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(base, wl, text_np)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
