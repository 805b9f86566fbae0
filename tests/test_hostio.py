"""Host I/O either side of the search path (bowtie_amd/csrc/bt_io.cpp through the C ABI): the C++
read parsers and output formatters against the unmodified reference's own outputs
(tests/golden/cli).  The search in the middle is done by the oracle here -- no GPU needed; the GPU
twin of these tests (test_gpu_cli.py) runs the bowtie-amd binary on the same cases."""
import os

import numpy as np
import pytest

import cli_cases as CC
import common as T
from bowtie_amd import hostio as H
from bowtie_amd.reads import pack_reads, parse_fastq


def run_case_paired(case, rd, pol, out, ex):
    oi = T.oracle_index(case["index"])
    spec = lambda x: ",".join(os.path.join(T.G, f) for f in x.split(","))
    b1 = H.read_all(spec(ex["mates1"]), mate=1, keep_raw=bool(case.get("dumps")), **rd)
    b2 = H.read_all(spec(ex["mates2"]), mate=2, keep_raw=bool(case.get("dumps")), **rd)
    ex["b2"] = b2
    # the insert limits as the aligner sees them: less the trimming at the fragment's outer ends (aligner.h:1921-1935)
    o1 = rd.get("trim5", 0) if pol.get("mate1_fw", True) else rd.get("trim3", 0)
    o2 = rd.get("trim3", 0) if pol.get("mate2_fw", False) else rd.get("trim5", 0)
    pol = dict(pol, min_ins=max(0, max(0, pol.get("min_ins", 0) - o1) - o2), max_ins=max(0, max(0, pol.get("max_ins", 250) - o1) - o2))
    cap = 2048 if pol.get("all_hits") else 2 * max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
    per = T.oracle_pair_results(case["index"], b1, b2, pol, cap=cap)
    hits, nh, st, pool = H.pack_hits(per, cap)
    opts = H.out_opts(**out)
    text, tally = H.format_pairs(b1, b2, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    ex["per_read"] = per
    return b1, text, tally, opts, ex, oi


def run_case(case):
    rd, pol, out, ex = CC.interpret(case["args"])
    if "mates1" in ex:
        return run_case_paired(case, rd, pol, out, ex)
    batch = H.read_all(CC.reads_spec(case), keep_raw=bool(case.get("dumps")), **rd)
    oi = T.oracle_index(case["index"])
    cap = 1024 if pol.get("all_hits") else max(pol.get("khits", 1), pol.get("mhits", 1) if pol.get("sample_max") else 1)
    per = T.oracle_results(case["index"], batch, pol, cap=cap)
    hits, nh, st, pool = H.pack_hits(per, cap)
    opts = H.out_opts(**out)
    text, tally = H.format_hits(batch, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    ex["per_read"] = per
    return batch, text, tally, opts, ex, oi


@pytest.mark.parametrize("case", CC.cases(), ids=lambda c: c["name"])
def test_parse_and_format_match_reference(case):
    want = CC.expected(case)
    batch, text, tally, opts, ex, oi = run_case(case)
    if opts.sam and not ex["sam_nohead"]:
        # the @PG line quotes the command line: take the reference's own
        head = [l for l in want.split(b"\n") if l.startswith(b"@")]
        cl = head[-1].split(b'CL:"', 1)[1][:-1].decode()
        text = H.sam_header(oi.refnames, oi.reflens, opts, cl, "\t".join(ex["rg"]) or None) + text
    if text != want:
        g, w = text.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(g, w)):
            assert a == b, "%s: line %d" % (case["name"], i)
        assert len(g) == len(w), case["name"]
    if case.get("dumps") and "b2" in ex:
        # pairs: first mates' records to <name>_1, second mates' to <name>_2; the ceiling counts mate alignments
        ceiling = 0xFFFFFFFF if opts.mhits == 0xFFFFFFFF else 2 * opts.mhits
        got = {k + t: b"" for k in ("AL", "UN", "MAX") for t in ("_1", "_2")}
        for r1, r2, (hs, tot, st) in zip(batch.raw, ex["b2"].raw, ex["per_read"]):
            k = "UN" if tot == 0 else ("AL" if tot <= ceiling else ("MAX" if "MAX_1" in case["dumps"] else "UN"))
            got[k + "_1"] += r1
            got[k + "_2"] += r2
        for k in case["dumps"]:
            assert got[k] == CC.expected_dump(case, k), (case["name"], k)
    elif case.get("dumps"):
        # --al / --un / --max: each read's record text goes to the file of its class (hit.h:385-488)
        mhits = opts.mhits
        got = {"AL": b"", "UN": b"", "MAX": b""}
        for raw, (hs, tot, st) in zip(batch.raw, ex["per_read"]):
            k = "UN" if tot == 0 else ("AL" if tot <= mhits else ("MAX" if "MAX" in case["dumps"] else "UN"))
            got[k] += raw
        for k in case["dumps"]:
            assert got[k] == CC.expected_dump(case, k), (case["name"], k)
    got_summary = H.summary(tally).strip().split("\n")
    assert got_summary == [l for l in case["stderr"] if l.startswith("#") or l.startswith("Reported") or l.startswith("No align")]


def test_fastq_parser_equals_python_reference_reader():
    b = H.read_all(os.path.join(T.G, "e_coli_1000.fq"), threads=3)
    p = pack_reads(parse_fastq(os.path.join(T.G, "e_coli_1000.fq")))
    assert b.n == p.n == 1000 and b.stride == p.stride
    assert (b.seq == p.seq).all() and (b.qual == p.qual).all() and (b.len == p.len).all()
    assert (b.seed == p.seed).all() and b.names == p.names


def test_batches_concatenate_and_ids_continue(tmp_path):
    """Small batches give the same reads as one large one; default names keep counting across batches and files."""
    spec = os.path.join(CC.D, "io.raw") + "," + os.path.join(CC.D, "io.raw")
    whole = H.read_all(spec, fmt="raw")
    parts = list(H.read_batches(spec, fmt="raw", max_reads=7))
    assert sum(x.n for x in parts) == whole.n == 80
    names = [nm for x in parts for nm in x.names]
    assert names == whole.names == [str(i).encode() for i in range(80)]
    assert np.concatenate([x.seed for x in parts]).tolist() == whole.seed.tolist()


@pytest.mark.parametrize("body,fmt,kw,msg", [
    (b"ACGT\n", "fastq", {}, "does not look like a FASTQ file"),
    (b"ACGT\n", "fasta", {}, "does not look like a FASTA file"),
    (b"@r\nACGT\n+\nII I\n", "fastq", {}, "Encountered a space"),
    (b"@r\nACGT\n+\nIII\n", "fastq", {}, "Too few quality values"),
    (b"@r\nACGT\n+\nIIIII\n", "fastq", {}, "more than 1024 quality values"),
    (b"@r\nACGT\n+\nII#\x1f\n", "fastq", {}, "expected 33-based Phred qual"),
    (b"@r\nACGT\n+\nII5I\n", "fastq", dict(quals="phred64"), "expected 64-based Phred qual"),
])
def test_malformed_input_reports_the_reference_message(tmp_path, body, fmt, kw, msg):
    p = tmp_path / "bad.txt"
    p.write_bytes(body)
    with pytest.raises(H.ReadInputError, match=msg):
        H.read_all(str(p), fmt=fmt, **kw)


def test_empty_inputs(tmp_path):
    p = tmp_path / "e.fa"
    p.write_bytes(b"")
    assert H.read_all(str(p), fmt="fasta") is None
    assert H.read_all(str(p), fmt="raw") is None


def _messy_fastq(rng, n):
    """FASTQ text with the things real files contain and a few they should not."""
    out = []
    for i in range(n):
        L = int(rng.integers(1, 60))
        seq = bytes(rng.choice(list(b"ACGTNacgtn"), size=L).tolist())
        qual = bytes(rng.integers(33, 74, size=L).tolist())
        name = b"rd%d" % i
        k = rng.integers(0, 14)
        if k == 0: name = b""
        elif k == 1: name += b" with words\tand tab"
        elif k == 2: seq = seq[:L // 2] + b"." + seq[L // 2 + 1:]
        elif k == 3: seq = seq[:L // 2] + b"-" + seq[L // 2:]          # dropped character: one quality too many -> error
        elif k == 4: seq = seq[:L // 2] + b"RYK"[:1] + seq[L // 2 + 1:]
        eol = b"\r\n" if k == 5 else b"\n"
        plus = b"+" + (name if k == 6 else b"")
        out.append(b"@" + name + eol + seq + eol + plus + eol + qual + eol)
    return out


@pytest.mark.parametrize("seed", range(6))
def test_fast_fastq_path_equals_step_by_step_parser(tmp_path, seed):
    """The bulk FASTQ path and the record-by-record parser that follows the reference must agree on
    every file: same reads, or the same error."""
    rng = np.random.default_rng(seed)
    recs = _messy_fastq(rng, 400)
    if seed % 2 == 0:
        recs = [r for r in recs if b"-" not in r.split(b"\n")[1]]       # an error-free file
    body = (b"\n\r\n" if seed % 3 == 0 else b"") + b"".join(recs)
    if seed % 3 == 1:
        body = body.rstrip(b"\r\n")                                        # no final newline
    p = tmp_path / "m.fq"
    p.write_bytes(body)
    kw = dict(trim5=int(seed % 3), trim3=int(seed % 2) * 2, seed=seed, skip=seed, upto=0 if seed < 3 else 300)

    def load(**extra):
        try:
            bs = list(H.read_batches(str(p), max_reads=97, threads=3, **kw, **extra))
        except H.ReadInputError as e:
            return ("error", str(e))
        return [(b.seq[i, :b.len[i]].tobytes(), b.qual[i, :b.len[i]].tobytes(), int(b.seed[i]), b.names[i])
                for b in bs for i in range(b.n)]
    fast, slow = load(), load(careful=True)
    assert fast == slow
    if seed % 2 == 0:
        assert isinstance(fast, list) and len(fast) > 50


_READER_DIGEST = r"""
import hashlib, sys
sys.path.insert(0, %r)
from bowtie_amd import hostio as H
spec, threads, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
h = hashlib.sha256()
n = 0
try:
    for b in H.read_batches(spec, max_reads=batch, threads=threads, seed=7):
        for i in range(b.n):
            h.update(b.seq[i, :b.len[i]].tobytes() + b"|" + b.qual[i, :b.len[i]].tobytes() + b"|" + b.names[i] + b"|%%d;" %% int(b.seed[i]))
        n += b.n
    print("ok", n, h.hexdigest())
except H.ReadInputError as e:
    print("error", n, str(e).replace(chr(10), " / "))
"""


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bulk_reader_is_the_same_whatever_the_fill_and_slice_sizes(tmp_path, seed):
    """The bulk FASTQ path reads an uncompressed file with several threads (a pread each) into the window and indexes the
    window's newlines with several threads; the light parse walks that index.  With the fills and the threads' slices
    shrunk to a few hundred bytes (BT_IO_FILL_BYTES / BT_IO_SLICE_BYTES: a record then straddles fills, slices, or both)
    the reads, names, seeds -- or the error -- are what the zlib path (BT_IO_NO_RAW=1, one thread's memchr until round 6)
    and the compressed copy of the files give; two files in one run, the first cut off inside a record."""
    import gzip
    import subprocess
    import sys
    rng = np.random.default_rng(100 + seed)
    recs = _messy_fastq(rng, 900)
    recs = [r for r in recs if b"-" not in r.split(b"\n")[1]]
    if seed == 1:
        recs[700] = b"@bad\nACGT\n+\nIIIIII\n"                              # more qualities than bases: the error must be the same
    a, b = tmp_path / "a.fq", tmp_path / "b.fq"
    a.write_bytes((b"\n\r\n" if seed == 2 else b"") + b"".join(recs[:500]) + (b"@cut\nACGT" if seed == 0 else b""))
    b.write_bytes(b"".join(recs[500:]).rstrip(b"\r\n") if seed == 2 else b"".join(recs[500:]))
    for f in (a, b):
        with gzip.open(str(f) + ".gz", "wb") as g:
            g.write(f.read_bytes())
    spec, spec_gz = "%s,%s" % (a, b), "%s.gz,%s.gz" % (a, b)
    code = _READER_DIGEST % T.ROOT

    def run(spec, threads, batch, **env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        p = subprocess.run([sys.executable, "-c", code, spec, str(threads), str(batch)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        return p.stdout.decode().strip()
    want = run(spec, 1, 97, BT_IO_NO_RAW=1)
    assert want.startswith("error" if seed == 1 else "ok")
    # (BT_IO_BULK_MIN: from how many whole records on the light parse lays them out in bulk, the threads a share each: 4096)
    for threads, batch, env in ((4, 97, {}), (4, 97, dict(BT_IO_FILL_BYTES=300, BT_IO_SLICE_BYTES=64)), (3, 1000, dict(BT_IO_FILL_BYTES=1500, BT_IO_SLICE_BYTES=100)),
                                (8, 5000, dict(BT_IO_FILL_BYTES=4096, BT_IO_SLICE_BYTES=16)), (4, 97, dict(BT_IO_NO_RAW=1, BT_IO_FILL_BYTES=300, BT_IO_SLICE_BYTES=64)),
                                (4, 97, dict(BT_IO_BULK_MIN=1)), (5, 97, dict(BT_IO_BULK_MIN=3, BT_IO_FILL_BYTES=700, BT_IO_SLICE_BYTES=64)),
                                (8, 5000, dict(BT_IO_BULK_MIN=1, BT_IO_FILL_BYTES=100000)), (3, 333, dict(BT_IO_BULK_MIN=2, BT_IO_NO_RAW=1, BT_IO_FILL_BYTES=2000)),
                                # BT_IO_POOL=0: every phase on threads of its own, as until round 6 (the default keeps them: par_for in bt_io.cpp)
                                (4, 97, dict(BT_IO_POOL=0, BT_IO_BULK_MIN=1)), (5, 1000, dict(BT_IO_POOL=0, BT_IO_FILL_BYTES=1500, BT_IO_SLICE_BYTES=100))):
        got = run(spec, threads, batch, **env)
        if batch == 97 or want.startswith("ok"):
            assert got == want, (threads, batch, env)
        else:
            assert got.split(" ", 2)[2] == want.split(" ", 2)[2], (threads, batch, env)       # the same error, whatever the batch it falls into
    assert run(spec_gz, 4, 97, BT_IO_FILL_BYTES=300, BT_IO_SLICE_BYTES=64) == want
    assert run(spec_gz, 4, 97, BT_IO_BULK_MIN=1) == want


def test_two_streams_parsed_at_once_from_two_threads(tmp_path):
    """The reader's phases run on a pool of threads that is kept (par_for, bt_io.cpp) and does one job at a time: a second stream
    parsed from another thread at the same moment starts threads of its own.  Both get what one alone gets."""
    import threading
    from bowtie_amd import hostio as H
    rng = np.random.default_rng(77)
    files, want = [], []
    for k in range(2):
        f = tmp_path / ("s%d.fq" % k)
        f.write_bytes(b"".join(r for r in _messy_fastq(rng, 12000) if b"-" not in r.split(b"\n")[1]))
        files.append(str(f))
        want.append([(b.n, b.seq.tobytes(), b.qual.tobytes(), b.len.tobytes(), list(b.names)) for b in H.read_batches(str(f), max_reads=5000, threads=1)])
    for _ in range(3):
        got = [None, None]

        def work(k):
            got[k] = [(b.n, b.seq.tobytes(), b.qual.tobytes(), b.len.tobytes(), list(b.names)) for b in H.read_batches(files[k], max_reads=5000, threads=4)]
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert got[0] == want[0] and got[1] == want[1]


def test_file_ending_inside_a_record_follows_the_reference(tmp_path):
    """Checked against the unmodified binary: with reads a, b and a cut-off third record the reference
    aligns only `a` -- its light parser gives up the record before a truncated one unless that record
    closed a 16-read batch (pat.cpp:822-858) -- and a stray blank line at the end counts as a cut-off record."""
    rec = lambda nm: b"@" + nm + b"\nACGTACGTAC\n+\nIIIIIIIIII\n"
    p = tmp_path / "t.fq"
    p.write_bytes(rec(b"a") + rec(b"b") + b"@c\nACG")
    assert H.read_all(str(p)).names == [b"a"]
    p.write_bytes(rec(b"a") + b"@c\nACG")
    assert H.read_all(str(p)) is None
    p.write_bytes(b"".join(rec(b"r%d" % i) for i in range(16)) + b"@c\nACG\n+")
    assert H.read_all(str(p)).n == 16                       # the 16th closed its batch (there the reference itself fails)
    p.write_bytes(b"".join(rec(b"r%d" % i) for i in range(5)) + b"\n")
    assert H.read_all(str(p)).n == 4
    p.write_bytes(b"".join(rec(b"r%d" % i) for i in range(5)) + b"@partial first line")
    assert H.read_all(str(p)).n == 5                        # EOF before the first newline: a clean end
    # the same when the cut falls exactly on our own batch boundary
    p.write_bytes(b"".join(rec(b"r%d" % i) for i in range(7)) + b"@c\nACG")
    assert [b.n for b in H.read_batches(str(p), max_reads=7)] == [6]
    # the dropped record's id goes to the next file's first read
    q = tmp_path / "u.raw"
    p.write_bytes(rec(b"") + rec(b"") + b"@c\nACG")
    names = [nm for b in H.read_batches(str(p) + "," + str(p)) for nm in b.names]
    assert names == [b"0", b"1"]
