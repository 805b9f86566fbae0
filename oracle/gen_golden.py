#!/usr/bin/env python3
"""Generate tests/golden/ -- TEST INFRASTRUCTURE, run in the build container (needs
/root/reference and `make -C oracle ref`).  Everything here comes from running the *unmodified*
reference binaries; the repo keeps only their data outputs:

  tests/golden/e_coli.{1,2,rev.1,rev.2}.ebwt   the reference's bundled index (data fixture)
  tests/golden/e_coli_1000.fq                   the reference's bundled reads
  tests/golden/multi.fa, multi.*.ebwt           small 5-sequence genome with N gaps (nFrag > 1),
                                                indexed by reference bowtie-build (--offrate 3)
  tests/golden/<index>__<reads>__<mode>.sam.gz  reference `bowtie -p 1 -S --sam-nohead` output with the
                                                SEQ and QUAL columns (test inputs) blanked to keep
                                                the fixtures small; the md5 of the *full* output is
                                                in MANIFEST.json and tests check both
  tests/golden/rank_vectors.json                mapLFEx / rowL / chase known answers (Appendix D)
  tests/golden/MANIFEST.json                    what was run, with md5s
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bowtie_amd.synth import synth_reads, write_fastq   # noqa: E402
from bowtie_amd.reads import parse_fastq, pack_reads     # noqa: E402
import oracle_lib as OL                                   # noqa: E402
from best_modes import BEST_CORE, BEST_EXTRA, PAIRED_MODES, PAIR_SETS   # noqa: E402
from bowtie_amd.synth import synth_pairs                 # noqa: E402

REF = "/root/reference"
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "oracle", "_ref")

MODES = {
    "v0": ["-v", "0"], "v1": ["-v", "1"], "v2": ["-v", "2"],
    "n0": ["-n", "0"], "n1": ["-n", "1"], "n2": ["-n", "2"], "n3": ["-n", "3"],
    "n2_k3": ["-n", "2", "-k", "3"], "v2_a": ["-v", "2", "-a"], "n2_m1": ["-n", "2", "-m", "1"],
    "n2_l20_e100": ["-n", "2", "-l", "20", "-e", "100"], "n2_nofw": ["-n", "2", "--nofw"],
    "v2_norc": ["-v", "2", "--norc"], "n2_nomaq": ["-n", "2", "--nomaqround"],
    "n3_l22_e140_k2": ["-n", "3", "-l", "22", "-e", "140", "-k", "2"],
    "v1_k5": ["-v", "1", "-k", "5"], "n1_a_m20": ["-n", "1", "-a", "-m", "20"],
}
# further option sets, run on three read sets only (EXTRA_SETS)
EXTRA_MODES = {
    "n3_a": ["-n", "3", "-a"], "n2_e200_nomaq": ["-n", "2", "-e", "200", "--nomaqround"],
    "n2_l12": ["-n", "2", "-l", "12"], "n2_maxbts10": ["-n", "2", "--maxbts", "10"], "n3_y": ["-n", "3", "-y"],
    "v2_k100": ["-v", "2", "-k", "100"], "n0_a_m5": ["-n", "0", "-a", "-m", "5"], "n1_l36_e40": ["-n", "1", "-l", "36", "-e", "40"],
    "v2_nofw_k3": ["-v", "2", "--nofw", "-k", "3"],
}
EXTRA_SETS = {("multi", "syn100"), ("multi", "syn50lowq"), ("e_coli", "syn100")}
MODES.update(EXTRA_MODES)
# the stateful best-first workers (tests/best_modes.py)
MODES.update({k: v[0] for k, v in BEST_CORE.items()})
MODES.update({k: v[0] for k, v in BEST_EXTRA.items()})
BEST_EXTRA_SETS = EXTRA_SETS | {("e_coli", "e_coli_1000")}
LONG_OK = ("n2", "v2", "n3", "v2_a", "n2_k3", "v0", "n1_a_m20", "n2_best", "v3", "n3_best", "v2_a_best_strata", "n2_M3")


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def make_multi():
    """5 sequences, 60 kbp total, with N gaps, a tandem repeat and a duplicated segment."""
    rng = np.random.default_rng(777)
    seqs = []
    dup = rng.integers(0, 4, size=1500)
    for k, L in enumerate([20000, 15000, 12000, 8000, 5000]):
        s = rng.integers(0, 4, size=L)
        if k in (0, 2):
            s[3000:4500] = dup                         # duplicated segment -> multi-mapping reads
        if k == 1:
            s[5000:5600] = np.tile(rng.integers(0, 4, size=6), 100)   # tandem repeat
        s = s.astype(np.int64)
        if k in (0, 1, 3):
            s[9000 % L:9000 % L + 37] = 4              # N gap -> fragment boundary
        if k == 0:
            s[15000:15010] = 4
        seqs.append(s)
    with open(os.path.join(G, "multi.fa"), "w") as f:
        for k, s in enumerate(seqs):
            f.write(">chrS%d synthetic %d\n" % (k + 1, k))
            txt = "".join("ACGTN"[c] for c in s)
            for i in range(0, len(txt), 70):
                f.write(txt[i:i + 70] + "\n")
    run([os.path.join(BIN, "bowtie-build-s"), "--offrate", "3", "--ftabchars", "6", "-q",
         os.path.join(G, "multi.fa"), os.path.join(G, "multi")])
    # multi.3.ebwt / .4.ebwt (the 2-bit reference, BitPairReference) stay: the paired-end path reads them


def strip_sam(sam: bytes) -> bytes:
    out = []
    for line in sam.split(b"\n"):
        if not line:
            continue
        c = line.split(b"\t")
        c[9] = b"*"; c[10] = b"*"
        out.append(b"\t".join(c))
    return b"\n".join(out) + (b"\n" if out else b"")


def main():
    os.makedirs(G, exist_ok=True)
    manifest = {"reference": "BenLangmead/bowtie v1.3.1", "runs": []}
    for ext in ("1.ebwt", "2.ebwt", "3.ebwt", "4.ebwt", "rev.1.ebwt", "rev.2.ebwt"):
        shutil.copyfile(os.path.join(REF, "indexes", "e_coli." + ext), os.path.join(G, "e_coli." + ext))
    shutil.copyfile(os.path.join(REF, "reads", "e_coli_1000.fq"), os.path.join(G, "e_coli_1000.fq"))
    make_multi()
    # synthetic read sets (their FASTQ is regenerated by tests from the same seeds; only the
    # reference's outputs are stored)
    sets = {}
    for idx in ("e_coli", "multi"):
        oi = OL.OracleIndex(os.path.join(G, idx))
        text = oi.joined_text()
        n = 400 if idx == "e_coli" else 600
        sets[(idx, "syn36")] = synth_reads(text, n, 36, mm_dist=(0, 0, 1, 1, 2, 3), seed=36)
        sets[(idx, "syn76")] = synth_reads(text, n, 76, mm_dist=(0, 0, 1, 1, 2, 3), seed=76)
        sets[(idx, "syn100")] = synth_reads(text, n, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=100)
        sets[(idx, "syn50lowq")] = synth_reads(text, n, 50, mm_dist=(0, 1, 2, 3, 5), seed=50, lowq_frac=0.15, n_frac=0.05)
        sets[(idx, "syn12")] = synth_reads(text, n // 2, 12, mm_dist=(0, 0, 1), seed=12, n_frac=0.03)
        # longer than the read-in-LDS builds take (register-window build), and the 105..112 band between the two
        sets[(idx, "syn150")] = synth_reads(text, n // 2, 150, mm_dist=(0, 1, 2, 3, 5), seed=150, lowq_frac=0.05, n_frac=0.02)
        sets[(idx, "syn110")] = synth_reads(text, n // 2, 110, mm_dist=(0, 1, 2, 3), seed=110)
    sets[("e_coli", "e_coli_1000")] = pack_reads(parse_fastq(os.path.join(G, "e_coli_1000.fq")))
    tmp = os.path.join(G, "_tmp.fq")
    for (idx, rname), batch in sets.items():
        write_fastq(batch, tmp)
        for mname, margs in MODES.items():
            if rname == "syn12" and "-a" in margs:
                continue  # 12-mers with -a: tens of thousands of hits per read, fixture too big
            if rname in ("syn150", "syn110") and mname not in LONG_OK:
                continue
            if mname in EXTRA_MODES and (idx, rname) not in EXTRA_SETS:
                continue
            if mname in BEST_EXTRA and (idx, rname) not in BEST_EXTRA_SETS:
                continue
            cmd = [os.path.join(BIN, "bowtie-align-s"), "--wrapper", "basic-0", "-p", "1", "-S", "--sam-nohead"] + \
                margs + ["-x", os.path.join(G, idx), tmp]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode != 0:
                manifest["runs"].append({"index": idx, "reads": rname, "mode": mname, "error": p.returncode})
                continue
            fn = "%s__%s__%s.sam.gz" % (idx, rname, mname)
            with gzip.GzipFile(os.path.join(G, fn), "wb", mtime=0) as f:
                f.write(strip_sam(p.stdout))
            manifest["runs"].append({"index": idx, "reads": rname, "mode": mname, "args": margs, "file": fn,
                                     "md5": hashlib.md5(p.stdout).hexdigest(),
                                     "summary": p.stderr.decode().strip().split("\n")})
    os.remove(tmp)
    # paired-end (-1/-2 --best: PairedBWAlignerV2)
    manifest["paired_runs"] = []
    for k in ("1", "2"):
        shutil.copyfile(os.path.join(REF, "reads", "e_coli_1000_%s.fq" % k), os.path.join(G, "e_coli_1000_%s.fq" % k))
    t1, t2 = os.path.join(G, "_tmp_1.fq"), os.path.join(G, "_tmp_2.fq")
    for idx, pname in PAIR_SETS:
        if pname == "e_coli_1000_pe":
            b1 = pack_reads(parse_fastq(os.path.join(G, "e_coli_1000_1.fq")))
            b2 = pack_reads(parse_fastq(os.path.join(G, "e_coli_1000_2.fq")))
        else:
            L = int(pname[2:])
            b1, b2 = synth_pairs(OL.OracleIndex(os.path.join(G, idx)).joined_text(), 400, L, seed=L)
        write_fastq(b1, t1)
        write_fastq(b2, t2)
        for mname, (margs, _) in PAIRED_MODES.items():
            cmd = [os.path.join(BIN, "bowtie-align-s"), "--wrapper", "basic-0", "-p", "1", "-S", "--sam-nohead"] + \
                margs + ["-x", os.path.join(G, idx), "-1", t1, "-2", t2]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode != 0:
                raise SystemExit("reference failed: " + " ".join(cmd))
            fn = "%s__%s__%s.sam.gz" % (idx, pname, mname)
            with gzip.GzipFile(os.path.join(G, fn), "wb", mtime=0) as f:
                f.write(strip_sam(p.stdout))
            manifest["paired_runs"].append({"index": idx, "reads": pname, "mode": mname, "args": margs, "file": fn,
                                            "md5": hashlib.md5(p.stdout).hexdigest(),
                                            "summary": p.stderr.decode().strip().split("\n")})
    os.remove(t1)
    os.remove(t2)
    # kernel-level known answers from the reference's own Ebwt methods are in SURVEY.md Appendix D
    # (micro-oracle that #includes ebwt.h); restated here as data
    vec = {"index": "e_coli", "fchr": [0, 1222723, 2474304, 3717743, 4938920], "zOff": 780711,
           "rows": {
               "0": {"L": 1, "lf": [0, 1222723, 2474304, 3717743]},
               "1": {"L": 3, "lf": [0, 1222724, 2474304, 3717743]},
               "223": {"L": 2, "lf": [26, 1222796, 2474357, 3717814]},
               "224": {"L": 1, "lf": [26, 1222796, 2474358, 3717814]},
               "225": {"L": 0, "lf": [26, 1222797, 2474358, 3717814]},
               "447": {"L": 0, "lf": [66, 1222844, 2474414, 3717893]},
               "448": {"L": 3, "lf": [67, 1222844, 2474414, 3717893]},
               "1000000": {"L": 2, "lf": [304523, 1516762, 2697082, 3896402]},
               "780711": {"L": 0, "lf": [242372, 1439386, 2647663, 3866060]},
               "780712": {"L": 1, "lf": [242372, 1439386, 2647663, 3866060]},
               "4938920": {"L": 1, "lf": [1222723, 2474303, 3717743, 4938920]}},
           "ftab": {"hi0": 0, "lo1": 826, "hi12345": 3739490, "lo12346": 3739837},
           "chase": {"row": 1000000, "joined": 3469571, "tidx": 0, "toff": 3469571}}
    with open(os.path.join(G, "rank_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)
    with open(os.path.join(G, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(manifest["runs"]), "runs,", len(manifest["paired_runs"]), "paired runs")


if __name__ == "__main__":
    main()
