#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS kernel's access pattern
(MI355X_MICROARCH.md: only wide streaming reads are calibrated; "calibrate on a known byte count in
your own access pattern").  bt_probe_rank_kernel does exactly the search kernel's index access -- per
row one 64-byte side as 4 x 16-byte loads plus the partner side's 8 counter bytes -- on N uniformly
random rows of the hg19-scale index, so the bytes it must move are known:
    reads : N x (64 B side + the 32-byte sector holding the partner counters) = 96 B, 128 B if the
            memory side works in 64-byte units; + 4 B/row for the row ids (streamed)
    writes: N x 17 B (lf[4] u32 + L u8, streamed)
Run under rocprofv3 by scripts/calib_fetch.sh; this script only does the launches."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
from bowtie_amd import _abi as A, aligner as AL        # noqa: E402
from bowtie_amd import ebwt_build as EB                # noqa: E402

N = int(os.environ.get("CALIB_ROWS", str(64 << 20)))
base, text, note = EB.ensure_big_index(0, torch.device("cuda", 0))
idx = AL.Index(base, need_mirror=False)
al = AL.Aligner(idx, A.make_policy(mode="v", mms=0))
rng = np.random.default_rng(1)
rows = rng.integers(0, len(text), size=N, dtype=np.int64).astype(np.uint32)
for _ in range(2):
    lf, L = al.probe_rank(rows)
print("probes per launch:", N, "index:", note)
