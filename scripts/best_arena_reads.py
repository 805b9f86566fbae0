#!/usr/bin/env python3
"""Where does bt_best_kernel read its per-lane arena?  Every lane of that kernel runs its own control flow, so each arena
word it fetches is a memory latency its whole wavefront waits through; this counts the fetches per function of
bt_best.h.  No GPU: a patched copy of the host build of the engine (tests/emu, under /tmp; the product header is not
touched) turns the arena accessor AW() into a counting proxy.  16-byte bt_ld4 loads of whole records are not counted.

  python scripts/best_arena_reads.py [--defines=-DBF_FAST_EXTEND=1] [--workload n2_best|pe] [--reads 3000]
"""
import argparse
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = "/tmp/bt_best_arena_reads"

PROXY = r'''#include <map>
#include <string>
static std::map<std::string, unsigned long long>& aw_tally() { static std::map<std::string, unsigned long long> m; return m; }
struct AwDump { AwDump() { aw_tally(); } ~AwDump() { unsigned long long t = 0; for (auto& kv : aw_tally()) t += kv.second; fprintf(stderr, "[aw] %-28s %12llu\n", "TOTAL", t); for (auto& kv : aw_tally()) fprintf(stderr, "[aw] %-28s %12llu  %5.1f%%\n", kv.first.c_str(), kv.second, 100.0 * kv.second / t); } };
static AwDump aw_dump;
struct AwProxy {
	uint32_t* p; const char* fn;
	operator uint32_t() const { aw_tally()[fn]++; return *p; }
	AwProxy& operator=(uint32_t v) { *p = v; return *this; }
	AwProxy& operator=(const AwProxy& o) { *p = (uint32_t)o; return *this; }
	AwProxy& operator|=(uint32_t v) { aw_tally()[fn]++; *p |= v; return *this; }
	AwProxy& operator&=(uint32_t v) { aw_tally()[fn]++; *p &= v; return *this; }
	AwProxy& operator+=(uint32_t v) { aw_tally()[fn]++; *p += v; return *this; }
};
#define AW(o) (AwProxy{&X.A[(o)], __func__})'''

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import emu_lib, common as T
from bowtie_amd import _abi as A
from bowtie_amd.synth import synth_reads, synth_pairs
emu_lib.LIB_PATH = %(lib)r
emu_lib.SRCS = []
em = emu_lib.EmuAligner(os.path.join(T.ROOT, "tests", "golden", "e_coli"))
text = T.joined_text("e_coli")
if %(workload)r == "pe":
    pol = A.make_policy(mode="n", mms=1, best=True, max_ins=500)
    b1, b2 = synth_pairs(text, %(reads)d // 2, 50, seed=7)[:2]
    res = em.align_pairs(pol, b1, b2, hit_cap=2)
    print("e_coli, %%d pairs of 50 bp, -n 1 --best: %%d aligned" %% (len(res), sum(1 for r in res if r[1] > 0)))
else:
    kw = T.MODES[%(workload)r]
    pol = A.make_policy(**kw)
    b = synth_reads(text, %(reads)d, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=99)
    res = em.align(pol, b, hit_cap=T.hit_cap_for(kw))
    print("e_coli, %%d reads of 100 bp, mode %%s: %%d aligned" %% (len(res), %(workload)r, sum(1 for r in res if r[1] > 0)))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--defines", default="")
    ap.add_argument("--workload", default="n2_best", help="a best-first mode of tests/common.py MODES, or 'pe' (paired -n 1 --best)")
    ap.add_argument("--reads", type=int, default=3000)
    a = ap.parse_args()
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W + "/tests")
    shutil.copytree(os.path.join(ROOT, "tests", "emu"), W + "/tests/emu", ignore=shutil.ignore_patterns("*.so"))
    shutil.copytree(os.path.join(ROOT, "bowtie_amd", "csrc"), W + "/bowtie_amd/csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), W + "/include")
    hp = W + "/bowtie_amd/csrc/bt_best.h"
    s = open(hp).read()
    assert s.count("#define AW(o) (X.A[(o)])") == 1
    open(hp, "w").write(s.replace("#define AW(o) (X.A[(o)])", PROXY))
    lib = W + "/libbt_emu_reads.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w"] + a.defines.split() +
                          ["-o", lib, W + "/tests/emu/bt_emu.cpp", W + "/bowtie_amd/csrc/bt_host.cpp"])
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib, workload=a.workload, reads=a.reads)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sys.stdout.write(p.stdout.decode())
    rows = sorted(set(l for l in p.stderr.decode().splitlines() if l.startswith("[aw]")), key=lambda l: -int(l.split()[2]))
    print("arena words fetched, by function (defines: %s)" % (a.defines or "none"))
    for l in rows[:24]:
        print(l[5:])
    if p.returncode != 0:
        sys.stderr.write(p.stderr.decode()[-2000:])
        sys.exit(1)


if __name__ == "__main__":
    main()
