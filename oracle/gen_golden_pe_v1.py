#!/usr/bin/env python3
"""Generate tests/golden/pe_v1/ -- TEST INFRASTRUCTURE, run in the build container (needs /root/reference and
`make -C oracle ref`).  Paired-end WITHOUT --best: the unmodified reference's `bowtie -p 1 -S --sam-nohead <options>
-1 .. -2 ..` (PairedBWAlignerV1, aligner.h:606-1480) on the same seeded pair sets the --best goldens use
(tests/common.py:pair_set), SEQ/QUAL blanked, md5 of the full output in MANIFEST.json."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bowtie_amd.synth import write_fastq                 # noqa: E402
import common as T                                       # noqa: E402
from best_modes import PAIRED_V1_MODES, PAIR_SETS        # noqa: E402
from gen_golden import strip_sam                         # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
D = os.path.join(G, "pe_v1")
BIN = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")


def main():
    os.makedirs(D, exist_ok=True)
    man = {"reference": "BenLangmead/bowtie v1.3.1, bowtie-align-s without --best (PairedBWAlignerV1)", "runs": []}
    tmpd = tempfile.mkdtemp()
    t1, t2 = os.path.join(tmpd, "r_1.fq"), os.path.join(tmpd, "r_2.fq")
    for idx, pname in PAIR_SETS:
        b1, b2 = T.pair_set(idx, pname)
        write_fastq(b1, t1)
        write_fastq(b2, t2)
        for mname, (margs, _) in PAIRED_V1_MODES.items():
            cmd = [BIN, "--wrapper", "basic-0", "-p", "1", "-S", "--sam-nohead"] + margs + \
                ["-x", os.path.join(G, idx), "-1", t1, "-2", t2]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode != 0:
                raise SystemExit("reference failed: %s\n%s" % (" ".join(cmd), p.stderr.decode()))
            fn = "pe_v1/%s__%s__%s.sam.gz" % (idx, pname, mname)
            with gzip.GzipFile(os.path.join(G, fn), "wb", mtime=0) as f:
                f.write(strip_sam(p.stdout))
            man["runs"].append({"index": idx, "reads": pname, "mode": mname, "args": margs, "file": fn,
                                "md5": hashlib.md5(p.stdout).hexdigest(),
                                "summary": p.stderr.decode().strip().split("\n")})
    with open(os.path.join(D, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1)
    print("wrote %d runs" % len(man["runs"]))


if __name__ == "__main__":
    main()
