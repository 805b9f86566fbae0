"""N > 1 path on CPU (gloo, world size 2): reads shard by rank with no data-path collective, and
the only exchange is the final counter all-reduce.  The per-rank search is played by the host
build of the automaton (tests/emu) since there is no GPU here; what is under test is the sharding
and the reduce -- exactly what bench.py does around bt_align_batch_device."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common as T


def shard(n, rank, world):
    per = (n + world - 1) // world
    return rank * per, min(n, (rank + 1) * per)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(T.ROOT, "tests"))
    import emu_lib as E
    from bowtie_amd import _abi as A
    from bowtie_amd.reads import ReadBatch
    batch = T.read_set("multi", "syn76")
    lo, hi = shard(batch.n, rank, world)
    mine = ReadBatch(batch.seq[lo:hi], batch.qual[lo:hi], batch.len[lo:hi], batch.seed[lo:hi], batch.names[lo:hi])
    res = E.EmuAligner(os.path.join(T.G, "multi")).align(A.make_policy(**T.MODES["n2"]), mine)
    aligned = sum(1 for h, _, _ in res if h)
    reported = sum(len(h) for h, _, _ in res)
    t = torch.tensor([aligned, reported, hi - lo, 0, 0], dtype=torch.int64)   # hit.h:169-175 counters
    dist.all_reduce(t)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), t.numpy())
    with open(os.path.join(out_dir, "d%d.txt" % rank), "w") as f:
        f.write(T.result_digest(res))
    dist.destroy_process_group()


def test_read_sharding_and_counter_reduce(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    batch = T.read_set("multi", "syn76")
    want = T.oracle_results("multi", batch, T.MODES["n2"])
    tot = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    assert (tot[0] == tot[1]).all()
    assert tot[0][0] == sum(1 for h, _, _ in want if h)
    assert tot[0][1] == sum(len(h) for h, _, _ in want)
    assert tot[0][2] == batch.n
    # concatenating the shards in rank order reproduces the unsharded result
    parts = []
    for r in range(world):
        lo, hi = shard(batch.n, r, world)
        parts.append(T.result_digest(want[lo:hi]))
        assert open(tmp_path / ("d%d.txt" % r)).read() == parts[-1]


def test_shard_covers_everything():
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 4, 8):
            spans = [shard(n, r, world) for r in range(world)]
            assert sum(max(0, b - a) for a, b in spans) == n
            assert all(spans[i][1] == spans[i + 1][0] or spans[i + 1][0] >= n for i in range(world - 1))


@pytest.mark.parametrize("how", ["self", "torchrun"])
def test_bench_gpus_2_starts_two_ranks(how):
    """`python bench.py --gpus 2 ...` on its own -- the shape of the driver's command -- starts two ranks itself; under
    torch.distributed.run it uses the ranks it is given.  --dry-ranks: rendezvous, the counter reduce with its all-gather
    check and rank 0's line, no GPU.  Counters: rank r contributes aligned 100+r, maxed r, reads 110+2r; wall 1+0.25r."""
    import json
    import socket
    import subprocess
    bench = os.path.join(T.ROOT, "bench.py")
    tail = ["--gpus", "2", "--steps", "2", "--dist-backend", "gloo", "--dry-ranks"]
    if how == "self":
        cmd = [sys.executable, bench] + tail
    else:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-800:]
    lines = [x for x in r.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1                               # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    assert d["config"]["hit_counters_last_step"]["aligned"] == 201 and d["config"]["hit_counters_last_step"]["maxed"] == 1
    assert d["config"]["reads"] == 222
    assert abs(d["value"] - 202 * 2 / 1.25) < 1e-9       # max over ranks of the wall time
    assert abs(d["reads_processed_per_s"] - 222 * 2 / 1.25) < 1e-9
