#!/usr/bin/env python3
"""Generate tests/golden/simple/ -- TEST INFRASTRUCTURE, run in the build container.

The reference's own known-answer suite, scripts/test/simple_tests.pl (101 cases, :42-890: tiny
genomes, every read format, -m, edits strings, paired-end geometry), re-expressed as data:

  * the @cases array is evaluated by perl and dumped as JSON (cases.json: the reference's test
    vectors, data only);
  * every distinct reference set is kept as FASTA (ref_<k>.fa); its index is built by the unmodified
    bowtie-build for the runs here and, byte-identically (checked below), by bowtie_amd/ebwt_build.py
    when the tests need it -- default bowtie-build parameters give a 4 MB ftab even for an 8-base
    genome, too much to store fifteen times;
  * every case the drop-in covers is run through the unmodified bowtie the way the perl harness
    runs it (`--quiet`, `-a` unless the case has its own report arguments; default output and
    -S), and stdout + exit status are stored (MANIFEST.json, <case>.<mode>.out.gz).
    Paired cases are run as written (PairedBWAlignerV1; bowtie-amd refuses those) and once more
    with --best appended (PairedBWAlignerV2), which is the variant the tests compare.
    Cases in --12 / --interleaved format are run as written and with --best appended too: with such input the
    reference always takes its stateful aligners, whether the records are pairs or not.
"""
import gzip
import hashlib
import json
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
D = os.path.join(ROOT, "tests", "golden", "simple")
BIN = os.path.join(ROOT, "oracle", "_ref")


def perl_cases():
    src = open(os.path.join(REF, "scripts", "test", "simple_tests.pl")).read()
    i = src.index("my @cases = (")
    j = src.index("\n);\n", i)
    prog = "use JSON::PP;\n" + src[i:j + 4] + "\nprint JSON::PP->new->canonical->encode(\\@cases);\n"
    out = subprocess.run(["perl", "-e", prog], stdout=subprocess.PIPE, check=True)
    return json.loads(out.stdout.decode())


def fastq_of(seqs, quals, names, mate=0):
    out = []
    for i, s in enumerate(seqs):
        q = (quals[i] if quals and i < len(quals) and quals[i] else "I" * len(s))
        nm = (names[i] if names and i < len(names) and names[i] else "r%d" % i)
        out.append("@%s%s\n%s\n+\n%s\n" % (nm, "/%d" % mate if mate else "", s, q))
    return "".join(out)


def inputs_of(c, k):
    """-> (format flag or None, {filename: content}, read arguments) the way runbowtie() sets them up."""
    fmt, files, rargs = None, {}, None
    base = "case%03d" % k
    for key, flag, ext in (("fastq", "-q", ".fq"), ("fasta", "-f", ".fa"), ("raw", "-r", ".raw")):
        if key in c:
            files[base + ext] = c[key]
            return flag, files, [os.path.join("simple", base + ext)]
        if key + "1" in c:
            files[base + ".1" + ext] = c[key + "1"]
            files[base + ".2" + ext] = c[key + "2"]
            return flag, files, ["-1", os.path.join("simple", base + ".1" + ext), "-2", os.path.join("simple", base + ".2" + ext)]
    if "cline_reads" in c:
        return "-c", files, [c["cline_reads"]]
    if "cline_reads1" in c:
        return "-c", files, ["-1", c["cline_reads1"], "-2", c["cline_reads2"]]
    if "cont_fasta_reads" in c:
        files[base + ".fa"] = c["cont_fasta_reads"]
        return None, files, [os.path.join("simple", base + ".fa")]
    if "tabbed" in c:
        files[base + ".tab"] = c["tabbed"]
        return None, files, ["--12", os.path.join("simple", base + ".tab")]
    if "interleaved" in c:
        files[base + ".il.fq"] = c["interleaved"]
        return None, files, ["--interleaved", os.path.join("simple", base + ".il.fq")]
    if "mate1s" in c:
        files[base + ".1.fq"] = fastq_of(c["mate1s"], c.get("qual1s"), c.get("names"), 1)
        files[base + ".2.fq"] = fastq_of(c["mate2s"], c.get("qual2s"), c.get("names"), 2)
        return "-q", files, ["-1", os.path.join("simple", base + ".1.fq"), "-2", os.path.join("simple", base + ".2.fq")]
    if "reads" in c:
        files[base + ".fq"] = fastq_of(c["reads"], c.get("quals"), c.get("names"))
        return "-q", files, [os.path.join("simple", base + ".fq")]
    return "skip", files, None


TMP = "/tmp/bt_simple_idx"


def same_as_own_builder(fa, ref_base):
    """bowtie_amd/ebwt_build.py must reproduce the reference index of this FASTA bit for bit (the tests build with it)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bowtie_amd import ebwt_build as EB
    from test_ebwt_build import read_fa
    names, seqs = read_fa(fa)
    EB.build_index(seqs, names, ref_base + "_own")
    for ext in ("1.ebwt", "2.ebwt", "3.ebwt", "4.ebwt", "rev.1.ebwt", "rev.2.ebwt"):
        a, b = open(ref_base + "." + ext, "rb").read(), open(ref_base + "_own." + ext, "rb").read()
        if a != b:
            raise SystemExit("own builder differs from bowtie-build on %s (%s)" % (fa, ext))
        os.remove(ref_base + "_own." + ext)


def main():
    os.makedirs(D, exist_ok=True)
    os.makedirs(TMP, exist_ok=True)
    cases = perl_cases()
    with open(os.path.join(D, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    G = os.path.dirname(D)
    refs = {}
    manifest = {"source": "scripts/test/simple_tests.pl:42-890 (BenLangmead/bowtie v1.3.1)", "cwd": "tests/golden", "cases": []}
    for k, c in enumerate(cases):
        key = json.dumps(c["ref"])
        if key not in refs:
            rid = len(refs)
            fa = os.path.join(D, "ref_%02d.fa" % rid)
            with open(fa, "w") as f:
                for i, s in enumerate(c["ref"]):
                    f.write(">%d\n%s\n" % (i, s))
            subprocess.run([os.path.join(BIN, "bowtie-build-s"), "-q", fa, os.path.join(TMP, "idx_%02d" % rid)], check=True,
                           stdout=subprocess.DEVNULL)
            same_as_own_builder(fa, os.path.join(TMP, "idx_%02d" % rid))
            refs[key] = rid
        rid = refs[key]
        fmt, files, rargs = inputs_of(c, k)
        entry = {"id": k, "name": c["name"], "ref": "simple/ref_%02d.fa" % rid, "should_abort": bool(c.get("should_abort"))}
        if fmt == "skip":
            entry["skipped"] = "--12 / --interleaved input is not in this build"
            manifest["cases"].append(entry)
            continue
        for fn, content in files.items():
            with open(os.path.join(D, fn), "w") as f:
                f.write(content)
        # `args` may be a list: the harness runs the case once per entry
        a0 = c.get("args", "")
        argsets = [shlex.split(x) for x in (a0 if isinstance(a0, list) else [a0])]
        tail = ["--quiet"] + shlex.split(c["report"] if "report" in c else "-a")
        # --12 / --interleaved: the file says whether its records are pairs; either way the reference then runs its
        # stateful aligners (ebwt_search.cpp:3001-3002), which bowtie-amd has for --best only
        one_file = rargs[0] in ("--12", "--interleaved")
        paired = rargs[0] == "-1" or (one_file and bool(c.get("paired")))
        entry.update({"reads": rargs, "paired": paired, "runs": []})
        if one_file:
            entry["needs_best"] = True
        for ai, aset in enumerate(argsets):
            args = ([fmt] if fmt else []) + aset + tail
            variants = [("asis", [])] + ([("best", ["--best"])] if (paired or one_file) and "--best" not in args else [])
            for vname, extra in variants:
                for mode, margs in (("default", []), ("sam", ["-S", "--sam-nohead"])):
                    cmd = [os.path.join(BIN, "bowtie-align-s"), "--wrapper", "basic-0", "-p", "1"] + args + extra + margs + \
                        ["-x", os.path.join(TMP, "idx_%02d" % rid)] + rargs
                    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=G)
                    fn = "simple/case%03d.a%d.%s.%s.out.gz" % (k, ai, vname, mode)
                    with gzip.GzipFile(os.path.join(G, fn), "wb", mtime=0) as f:
                        f.write(p.stdout)
                    entry["runs"].append({"variant": vname, "args": args + extra + margs, "returncode": p.returncode, "file": fn,
                                          "md5": hashlib.md5(p.stdout).hexdigest(),
                                          "stderr": p.stderr.decode(errors="replace").strip().split("\n")[:6]})
        manifest["cases"].append(entry)
        print(k, c["name"], [r["returncode"] for r in entry["runs"]])
    with open(os.path.join(D, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(len(refs), "distinct references;", sum(1 for e in manifest["cases"] if "skipped" in e), "cases skipped")


if __name__ == "__main__":
    main()
