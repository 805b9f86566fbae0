#!/usr/bin/env python3
"""Per-source-line static instruction counts of one bt_search_kernel instance (companion of isa_by_state.py).
  python scripts/isa_by_line.py [--instance Li3ELb0ELb1ELb1E] [--file bt_core.h] [--min 4] [--asm /tmp/x.s (reuse)]
"""
import argparse, collections, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bowtie_amd", "csrc")
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instance", default="Li3ELb0ELb1ELb1E")
    ap.add_argument("--file", default="")
    ap.add_argument("--min", type=int, default=4)
    ap.add_argument("--asm", default="")
    ap.add_argument("--kernel", default="_Z16bt_search_kernelI")
    ap.add_argument("--src", default="bt_kernels.hip")
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    out = a.asm or "/tmp/bt_kernels_lines.s"
    if not a.asm or not os.path.exists(out):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-gline-tables-only"] + a.flags.split() +
                              ["-o", out, os.path.join(CS, a.src)], stderr=subprocess.DEVNULL, cwd=CS)
    s = open(out).read()
    m = re.search(r"^(" + a.kernel + a.instance + r"\w*):[^\n]*\n", s, re.M)
    body = s[m.end():s.index(".end_amdhsa_kernel", m.end())].splitlines()
    cnt = collections.defaultdict(lambda: [0, 0, 0])
    cur = ("?", 0)
    for l in body:
        t = l.strip()
        mm = re.match(r"\.loc\s+\d+\s+\d+.*;\s*(\S+?):(\d+):", t)
        if mm:
            if int(mm.group(2)) > 0: cur = (os.path.basename(mm.group(1)), int(mm.group(2)))
            continue
        if not t or t[0] in ";." or t.endswith(":"): continue
        op = t.split()[0]
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2
        cnt[cur][k] += 1
    srcs = {}
    tot = [0, 0, 0]
    for (f, ln), v in sorted(cnt.items()):
        for k in range(3): tot[k] += v[k]
        if a.file and f != a.file: continue
        if v[0] + v[1] + v[2] < a.min: continue
        if f not in srcs:
            p = os.path.join(CS, f)
            srcs[f] = open(p).read().splitlines() if os.path.exists(p) else []
        text = srcs[f][ln - 1].strip()[:110] if 0 < ln <= len(srcs[f]) else ""
        print("%-16s %5d  V%4d S%4d M%3d  %s" % (f, ln, v[0], v[1], v[2], text))
    print("total V%d S%d M%d" % tuple(tot))
    mm = re.search(r"\.vgpr_count:\s+(\d+)", s[m.end():]); 
    for key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        pass
main()
