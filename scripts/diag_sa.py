"""Diagnostics for bowtie_amd/ebwt_build.py at scale (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bowtie_amd import ebwt_build as EB

dev = torch.device("cuda")
for n in [int(x) for x in sys.argv[1:]] or [300_000_000, 1_200_000_000, 2_860_000_000]:
    torch.cuda.empty_cache()
    g = torch.Generator(device=dev); g.manual_seed(7)
    s = torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.uint8)
    if os.environ.get("DIAG_REPEATS"):
        s = EB.synth_genome(n, dev)[0]
        n = s.numel()
    t0 = time.time()
    key = EB._kmer32_keys(s)
    # 1. keys vs direct computation on a sample
    i = torch.randint(0, n - 40, (1_000_000,), generator=g, device=dev)
    direct = torch.zeros_like(i)
    for k in range(32):
        direct = (direct << 2) | s[i + k].to(torch.int64)
    direct ^= -(1 << 63)
    print(n, "key mismatches in sample:", int((key[i] != direct).sum()), flush=True)
    # 2. bucket extraction + sort sanity
    tot = 0
    for b in range(16):
        idx = EB._nonzero((((key >> 60) & 15) ^ 8) == b)
        kk = key[idx]
        wrong_bucket = int(((((kk >> 60) & 15) ^ 8) != b).sum())
        ks, perm = torch.sort(kk)
        unsorted = int((ks[1:] < ks[:-1]).sum())
        perm_ok = int((kk[perm] != ks).sum())
        tot += idx.numel()
        if wrong_bucket or unsorted or perm_ok or b in (0, 15):
            print("  bucket", b, idx.numel(), "wrong_bucket", wrong_bucket, "unsorted", unsorted, "perm_mismatch", perm_ok, flush=True)
        del idx, kk, ks, perm
    print("  total", tot, "expected", n + 1, flush=True)
    del key
    sa = EB.suffix_array(s)
    print("  SA bad pairs:", EB.check_sa_sample(s, sa), " perm check:", int(torch.bincount((sa % 1000003), minlength=1).numel()), "time %.1fs" % (time.time() - t0), flush=True)
    # is sa a permutation?  sum check
    print("  sum(sa) ok:", int(sa.sum().item()) == n * (n + 1) // 2, flush=True)
    del sa, s
