#!/usr/bin/env python3
"""Static instruction counts of bt_search_kernel by the source region they come from: the kernel wrapper's request-issue and
rank section, each state block of the automaton (bt_core.h), the helpers.  Compiles the device code with line tables and
attributes every ISA instruction to its .loc.  No GPU.

  python scripts/isa_by_state.py [--instance Li3ELb0ELb1ELb1E]
"""
import argparse
import collections
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "bowtie_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instance", default="Li3ELb0ELb1ELb1E", help="template instance: OCC, EXT, RL, LITE")
    a = ap.parse_args()
    out = "/tmp/bt_kernels_lines.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-gline-tables-only",
                           "-o", out, os.path.join(CS, "bt_kernels.hip")], stderr=subprocess.DEVNULL, cwd=CS)
    s = open(out).read()
    files = {int(m.group(1)): m.group(2) for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s, re.M)}
    m = re.search(r"^(_Z16bt_search_kernelI" + a.instance + r"\w+):[^\n]*\n", s, re.M)
    body = s[m.end():s.index(".end_amdhsa_kernel", m.end())].splitlines()
    # state blocks of bt_core.h by line range
    core = open(os.path.join(CS, "bt_core.h")).read().splitlines()
    marks = []
    for i, l in enumerate(core, 1):
        mm = re.search(r"if \(ST_IS(?:_NOREQ)?\((ST_\w+)\)", l)
        if mm:
            marks.append((i, mm.group(1)))
        for pat, nm in (("if (L.state == ST_STEP_LFDONE || L.state == ST_STEP_POST)", "ST_STEP_LFDONE/POST"), ("if (L.state == ST_STEP_BEGIN) {", "ST_STEP_BEGIN"),
                        ("if (L.state == ST_CHASE_CHECK) {", "ST_CHASE_CHECK"), ("if (L.state == ST_CHASE_LFDONE) {", "ST_CHASE_LFDONE"),
                        ("BT_HD void bt_lane_run(", "(bt_lane_run head)"), ("BT_HD void bt_lane_slow(", "(bt_lane_slow head)"),
                        ("BT_HD void bt_lane_start(", "bt_lane_start"), ("BT_HD void bt_rescan_piece(", "ST_RESCAN (bt_rescan_piece)"), ("BT_HD void bt_candscan_piece(", "ST_CANDSCAN (bt_candscan_piece)"), ("BT_HD bool bt_report_hit(", "bt_report_hit"), ("BT_HD void bt_lane_finish(", "bt_lane_finish")):
            if pat in l:
                marks.append((i, nm))
    marks.sort()

    def region(fname, line):
        if fname == "bt_core.h":
            nm = "(bt_core.h helpers)"
            for ln, n in marks:
                if ln <= line:
                    nm = n
                else:
                    break
            return nm
        if fname == "bt_kernels.hip":
            return "kernel: request issue, loop, tallies"
        if fname == "bt_rank.h":
            return "rank arithmetic (bt_rank.h)"
        return "(runtime headers)"
    cnt = collections.defaultdict(lambda: [0, 0, 0])
    cur = ("bt_kernels.hip", 0)
    for l in body:
        t = l.strip()
        mm = re.match(r"\.loc\s+\d+\s+\d+.*;\s*(\S+?):(\d+):", t)
        if mm:
            if int(mm.group(2)) > 0:
                cur = (os.path.basename(mm.group(1)), int(mm.group(2)))
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2
        cnt[region(*cur)][k] += 1
    tot = [sum(v[k] for v in cnt.values()) for k in range(3)]
    print("%-40s %6s %6s %6s" % ("region (" + a.instance + ")", "VALU", "SALU", "mem/LDS"))
    for nm, v in sorted(cnt.items(), key=lambda x: -x[1][0]):
        print("%-40s %6d %6d %6d" % (nm, v[0], v[1], v[2]))
    print("%-40s %6d %6d %6d" % ("total", tot[0], tot[1], tot[2]))


if __name__ == "__main__":
    main()
