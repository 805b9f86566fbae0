#!/usr/bin/env python3
"""Generate tests/golden/cli/ -- TEST INFRASTRUCTURE, run in the build container (needs
/root/reference and `make -C oracle ref`).  Pins the host I/O surface either side of the search
path: read-file parsing (FASTQ / FASTA / raw / -c, trimming, -s/-u, quality
encodings, gz, several files) and the reference's two output formats with their options.

Every expected output here is the stdout/stderr of the *unmodified* reference binary
(oracle/_ref/bowtie-align-s -p 1); the repo keeps the small input files made below, the outputs
(gzip) and MANIFEST.json.  Paths in the commands are relative to tests/golden/.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bowtie_amd.synth import synth_reads, synth_pairs, write_fastq   # noqa: E402
import oracle_lib as OL                                   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
D = os.path.join(G, "cli")
BIN = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-s")


def fastq_records(path, n):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return [lines[4 * i:4 * i + 4] for i in range(n)]


def make_inputs():
    os.makedirs(D, exist_ok=True)
    recs = fastq_records(os.path.join(G, "e_coli_1000.fq"), 120)
    # edge cases of the FASTQ parser, each on a read that still aligns somewhere
    recs[3][0] = b"@r3 extra words\tand a tab"
    recs[5][0] = b"@"                                             # empty name -> the read id
    recs[7][1] = recs[7][1][:10].lower() + b"." + recs[7][1][11:]   # lower case, '.' -> N
    recs[9][1] = recs[9][1][:5] + b"-*" + recs[9][1][5:]          # non-alphabetic characters are dropped
    recs[11][1] = recs[11][1][:20] + b"RYK" + recs[11][1][23:]    # IUPAC codes -> N
    recs[13] = [x + b"\r" for x in recs[13]]                      # CRLF record
    recs[14][2] = b"+" + recs[14][0][1:]                          # '+' line repeats the name
    body = b"\n\n" + b"\n".join(b"\n".join(r) for r in recs)       # leading blank lines; no final newline
    with open(os.path.join(D, "io.fq"), "wb") as f:
        f.write(body)
    with gzip.GzipFile(os.path.join(D, "io.fq.gz"), "wb", mtime=0) as f:
        f.write(b"\n".join(b"\n".join(r) for r in recs[:40]) + b"\n")
    # phred+64 and solexa-scaled copies of the first 50 reads
    plain = fastq_records(os.path.join(G, "e_coli_1000.fq"), 50)
    with open(os.path.join(D, "io64.fq"), "wb") as f:
        for r in plain:
            f.write(b"\n".join([r[0], r[1], r[2], bytes(q + 31 for q in r[3])]) + b"\n")
    rng = np.random.default_rng(64)
    with open(os.path.join(D, "iosol.fq"), "wb") as f:
        for r in plain:
            sol = rng.integers(-5, 41, size=len(r[1]))
            f.write(b"\n".join([r[0], r[1], r[2], bytes(int(64 + s) for s in sol)]) + b"\n")
    # integer qualities (Phred, and Solexa-scaled with negative values), space-separated
    with open(os.path.join(D, "ioint.fq"), "wb") as f:
        for r in plain[:40]:
            f.write(b"\n".join([r[0], r[1], r[2], b" ".join(str(q - 33).encode() for q in r[3])]) + b"\n")
    with open(os.path.join(D, "iointsol.fq"), "wb") as f:
        for r in plain[:40]:
            sol = rng.integers(-5, 41, size=len(r[1]))
            f.write(b"\n".join([r[0], r[1], r[2], b" ".join(str(int(x)).encode() for x in sol)]) + b"\n")
    # a two-record FASTA for -F: multi-line sequence, ambiguity codes, '-', lower case, a description after the name
    oi_e = OL.OracleIndex(os.path.join(G, "e_coli"))
    et = "".join("ACGT"[c] for c in oi_e.joined_text()[100000:100900])
    with open(os.path.join(D, "cont.fa"), "wb") as f:
        a_ = et[:400]
        a_ = a_[:90] + "RY-N" + a_[94:200].lower() + a_[200:]
        f.write(b">ctgA first contig\n")
        for i in range(0, len(a_), 60):
            f.write(a_[i:i + 60].encode() + b"\n")
        f.write(b">ctgB\n" + et[400:900].encode())
    # FASTA: one-line records, a two-line record (only its first line is read), an empty name,
    # blank lines after a name, and a last record without a final newline (loses its last base)
    with open(os.path.join(D, "io.fa"), "wb") as f:
        f.write(b"\n")
        for i, r in enumerate(plain[:40]):
            name = b"" if i == 4 else r[0][1:] + (b" desc" if i == 2 else b"")
            seq = r[1]
            if i == 6:
                seq = seq[:18] + b"\n" + seq[18:]
            if i == 8:
                f.write(b">" + name + b"\n\n" + seq + b"\n")
                continue
            f.write(b">" + name + b"\n" + seq + (b"" if i == 39 else b"\n"))
    # raw: one sequence per line, blank lines, a CRLF line
    with open(os.path.join(D, "io.raw"), "wb") as f:
        f.write(b"\n")
        for i, r in enumerate(plain[:40]):
            f.write(r[1] + (b"\r\n" if i == 3 else b"\n") + (b"\n" if i == 5 else b""))
    # reads for the multi-sequence index (names with a space; repeats -> many hits)
    oi = OL.OracleIndex(os.path.join(G, "multi"))
    text = oi.joined_text()
    b = synth_reads(text, 150, 50, mm_dist=(0, 0, 1, 2), seed=515)
    b.names = [nm + (b" lane 3" if i % 7 == 0 else b"") for i, nm in enumerate(b.names)]
    write_fastq(b, os.path.join(D, "multi.fq"))
    # read pairs for the multi-sequence index: most names end in /1 and /2, some are bare (the reference
    # appends the suffix), one has a description, one pair has a 3-base mate (skipped with a warning)
    p1, p2 = synth_pairs(text, 150, 40, frag_lo=120, frag_hi=300, seed=4040)
    for i in range(150):
        if i % 11 == 3:
            p1.names[i] = p1.names[i][:-2]; p2.names[i] = p2.names[i][:-2]
        if i == 20:
            p1.names[i] = b"frag20 lane 3/1"; p2.names[i] = b"frag20 lane 3/2"
    p2.len[57] = 3
    write_fastq(p1, os.path.join(D, "pe_1.fq"))
    write_fastq(p2, os.path.join(D, "pe_2.fq"))
    e1, e2 = synth_pairs(oi_e.joined_text(), 120, 50, seed=5151)
    write_fastq(e1, os.path.join(D, "pee_1.fq"))
    write_fastq(e2, os.path.join(D, "pee_2.fq"))
    return plain


def cases(plain):
    cseq = ",".join([plain[0][1].decode(), plain[1][1].decode() + ":" + plain[1][3].decode(),
                     plain[2][1][:30].decode(), "ACGTTGCANNACGT" + plain[3][1][:20].decode()])
    E, M = "e_coli", "multi"
    return [
        ("fq_default", E, ["-n", "2"], "cli/io.fq"),
        ("fq_sam_head", E, ["-S", "-n", "2"], "cli/io.fq"),
        ("fq_v2_a_suppress", E, ["-v", "2", "-a", "--suppress", "1,5,6"], "cli/io.fq"),
        ("fq_refidx_B1_k3", E, ["-v", "1", "-k", "3", "--refidx", "-B", "1"], "cli/io.fq"),
        ("fq_cost_showseed_seed", E, ["-n", "1", "--cost", "--showseed", "--seed", "7"], "cli/io.fq"),
        ("fq_trim", E, ["-5", "3", "-3", "2", "-v", "2"], "cli/io.fq"),
        ("fq_trim_sam", E, ["-5", "1", "-3", "4", "-n", "2", "-S", "--sam-nohead"], "cli/io.fq"),
        ("fq_skip_upto", E, ["-s", "10", "-u", "50", "-S", "--sam-nohead"], "cli/io.fq"),
        ("fq_gz_two_files", E, ["-v", "0", "-S", "--sam-nohead"], "cli/io.fq.gz,cli/io.fq"),
        ("fq_sam_opts", E, ["-S", "--sam-RG", "ID:grp1", "--sam-RG", "SM:x", "--mapq", "30", "--no-unal"], "cli/io.fq"),
        ("fq_sam_nosq_refidx", E, ["-S", "--sam-nosq", "--refidx", "-v", "1"], "cli/io.fq"),
        ("fq_tryhard_nofw", E, ["-y", "--nofw", "-n", "3", "-e", "120"], "cli/io.fq"),
        ("fq_maxbts_norc_offrate", E, ["--maxbts", "20", "--norc", "-o", "7", "-n", "2", "-l", "20"], "cli/io.fq"),
        ("fa_default", E, ["-f", "-v", "2"], "cli/io.fa"),
        ("fa_sam", E, ["-f", "-S", "--sam-nohead", "-n", "3"], "cli/io.fa"),
        ("fa_trim", E, ["-f", "-5", "2", "-3", "3", "-v", "1", "-S", "--sam-nohead"], "cli/io.fa"),
        ("raw_default", E, ["-r", "-v", "1"], "cli/io.raw"),
        ("raw_sam", E, ["-r", "-S", "--sam-nohead", "-s", "2"], "cli/io.raw"),
        ("cmdline", E, ["-c", "-v", "2"], cseq),
        ("cmdline_sam_trim", E, ["-c", "-S", "--sam-nohead", "-5", "2", "-n", "2"], cseq),
        ("phred64", E, ["--phred64-quals", "-n", "2", "-S", "--sam-nohead"], "cli/io64.fq"),
        ("solexa", E, ["--solexa-quals", "-n", "2", "-S", "--sam-nohead"], "cli/iosol.fq"),
        ("multi_default_fullref", M, ["--fullref", "-v", "2", "-k", "4"], "cli/multi.fq"),
        ("multi_sam_notrunc", M, ["-S", "--sam-no-qname-trunc", "-n", "2", "-k", "2"], "cli/multi.fq"),
        ("multi_sam_fullref", M, ["-S", "--fullref", "-v", "1"], "cli/multi.fq"),
        ("multi_all_m3", M, ["-a", "-m", "3", "-v", "2", "-S", "--sam-nohead"], "cli/multi.fq"),
        ("multi_all", M, ["-a", "-v", "2"], "cli/multi.fq"),
        ("multi_k2_m5", M, ["-k", "2", "-m", "5", "-n", "1"], "cli/multi.fq"),
        ("fasta_cont", E, ["-F", "40,13", "-v", "2", "-S", "--sam-nohead"], "cli/cont.fa"),
        ("fasta_cont_trim_dump", E, ["-F", "50,7", "-5", "3", "-3", "2", "-n", "2", "-s", "4", "--al", "AL", "--un", "UN"], "cli/cont.fa,cli/cont.fa"),
        ("intquals", E, ["--integer-quals", "-n", "2", "-S", "--sam-nohead"], "cli/ioint.fq"),
        ("intquals_solexa", E, ["--integer-quals", "--solexa-quals", "-n", "2", "-S", "--sam-nohead"], "cli/iointsol.fq"),
        ("intquals_trim5", E, ["--integer-quals", "-5", "2", "-v", "2"], "cli/ioint.fq"),
        # --al / --un / --max: AL, UN, MAX stand for the dump files (their contents are stored too)
        ("dump_multi_m3", M, ["-a", "-m", "3", "-v", "2", "--al", "AL", "--un", "UN", "--max", "MAX"], "cli/multi.fq"),
        ("dump_multi_nomax", M, ["-k", "2", "-m", "5", "-n", "1", "--al", "AL", "--un", "UN"], "cli/multi.fq"),
        ("dump_fq", E, ["-n", "2", "--un", "UN", "--al", "AL"], "cli/io.fq"),
        ("dump_fq_trim_skip", E, ["-5", "3", "-3", "2", "-s", "5", "-u", "40", "--al", "AL", "--un", "UN"], "cli/io.fq"),
        ("dump_fa", E, ["-f", "-v", "2", "--al", "AL", "--un", "UN"], "cli/io.fa"),
        ("dump_raw", E, ["-r", "-v", "1", "--un", "UN"], "cli/io.raw"),
        ("dump_cmdline", E, ["-c", "-v", "2", "--al", "AL", "--un", "UN"], cseq),
        # the stateful best-first workers: --best, --strata, -M, -v 3
        ("best_default", E, ["--best"], "cli/io.fq"),
        ("best_sam_n2", E, ["--best", "-S", "--sam-nohead", "-n", "2"], "cli/io.fq"),
        ("v3_default", E, ["-v", "3"], "cli/io.fq"),
        ("best_strata_a", M, ["-a", "--best", "--strata", "-v", "2"], "cli/multi.fq"),
        ("best_strata_k3_m5_sam", M, ["-k", "3", "--best", "--strata", "-m", "5", "-S", "--sam-nohead"], "cli/multi.fq"),
        ("bigM_verbose", M, ["-M", "2", "-v", "2"], "cli/multi.fq"),
        ("bigM_sam_dump", M, ["-M", "3", "--best", "-n", "2", "-S", "--sam-nohead", "--max", "MAX", "--un", "UN", "--al", "AL"], "cli/multi.fq"),
        ("bigM_k2_cost", M, ["-M", "4", "-k", "2", "--best", "--cost", "-v", "3"], "cli/multi.fq"),
        ("best_maxbts_nofw", E, ["--best", "--maxbts", "3", "--nofw", "-n", "3", "-e", "100"], "cli/io.fq"),
        # reads trimmed to fewer than 4 bases, or to nothing: skipped with a warning, counted as unaligned
        ("best_trim_short", E, ["--best", "-3", "32", "-n", "2"], "cli/io.fq"),
        ("n2_trim_short_sam", E, ["-3", "33", "-n", "2", "-S", "--sam-nohead"], "cli/io.fq"),
        ("n2_trim_away", E, ["-3", "40", "-n", "2", "-S", "--sam-nohead"], "cli/io.fq"),
        ("best_trim_away", E, ["-5", "20", "-3", "20", "--best", "-v", "1"], "cli/io.fq"),
        # paired-end (-1/-2 --best: PairedBWAlignerV2); no <s> argument
        ("pe_default", M, ["--best", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_sam_head", M, ["--best", "-S", "-X", "400", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_n1_k3_sam", M, ["--best", "-n", "1", "-k", "3", "-X", "350", "-S", "--sam-nohead", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_a_strata_cost", M, ["--best", "--strata", "-a", "-v", "2", "-X", "400", "--cost", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_m1_I100", M, ["--best", "-m", "1", "-I", "100", "-X", "300", "-S", "--sam-nohead", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_dump_m1", M, ["--best", "-m", "1", "-X", "400", "--al", "AL", "--un", "UN", "--max", "MAX", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_dump_M1_nomax_sam", M, ["--best", "-M", "1", "-X", "400", "-S", "--sam-nohead", "--al", "AL", "--un", "UN", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_M1_sam", M, ["--best", "-M", "1", "-X", "400", "-S", "--sam-nohead", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_M2_strata_default", M, ["--best", "--strata", "-M", "2", "-v", "2", "-X", "400", "--cost", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_trim_ff", M, ["--best", "--ff", "-5", "2", "-3", "3", "-X", "300", "-1", "cli/pe_1.fq", "-2", "cli/pe_2.fq"], ""),
        ("pe_ecoli_config5", E, ["-n", "1", "--best", "-X", "500", "-S", "--sam-nohead", "--no-unal", "-1", "cli/pee_1.fq", "-2", "cli/pee_2.fq"], ""),
        ("pe_ecoli_two_files", E, ["-v", "2", "--best", "-X", "500", "-1", "cli/pee_1.fq,cli/pe_1.fq", "-2", "cli/pee_2.fq,cli/pe_2.fq"], ""),
    ]


def main():
    plain = make_inputs()
    manifest = {"reference": "BenLangmead/bowtie v1.3.1", "cwd": "tests/golden", "cases": []}
    for name, idx, args, reads in cases(plain):
        dump_paths = {k: os.path.join(D, "_dump_%s.txt" % k) for k in ("AL", "UN", "MAX") if k in args}
        cmd = [BIN, "--wrapper", "basic-0", "-p", "1"] + [dump_paths.get(a, a) for a in args] + ["-x", idx] + ([reads] if reads else [])
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=G)
        dumps = {}
        paired_dumps = "-1" in args and bool(dump_paths)        # pairs: <name>_1.txt and <name>_2.txt (hit.h:629-660)
        for k, path in dump_paths.items():
            for tag in (("_1", "_2") if paired_dumps else ("",)):
                src = path[:-4] + tag + ".txt"
                data = b""
                if os.path.exists(src):
                    with open(src, "rb") as f:
                        data = f.read()
                    os.remove(src)
                fn = "cli/%s.%s%s.gz" % (name, k.lower(), tag)
                with gzip.GzipFile(os.path.join(G, fn), "wb", mtime=0) as f:
                    f.write(data)
                dumps[k + tag] = fn
        entry = {"name": name, "index": idx, "args": args, "reads": reads, "returncode": p.returncode, "dumps": dumps,
                 "stderr": p.stderr.decode(errors="replace").strip().split("\n"),
                 "md5": hashlib.md5(p.stdout).hexdigest(), "file": "cli/%s.out.gz" % name}
        with gzip.GzipFile(os.path.join(G, entry["file"]), "wb", mtime=0) as f:
            f.write(p.stdout)
        manifest["cases"].append(entry)
        print(name, p.returncode, len(p.stdout), entry["stderr"][-1])
    with open(os.path.join(D, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
