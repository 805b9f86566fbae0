"""The reference's own known-answer suite (scripts/test/simple_tests.pl:42-890: 8-base genomes,
ftabChars > genome length, every read format, -m, trimming, edits strings, paired-end geometry incl.
--allow-contain) as data: tests/golden/simple (oracle/gen_simple_tests.py ran every case through the
unmodified reference binary; paired cases also with --best, the aligner this build has).

CPU: C++ read parsers -> oracle search -> C++ formatters == the reference's stdout.
GPU (-m gpu): the bowtie-amd binary on the same command lines == the reference's stdout / exit status."""
import gzip
import json
import os
import subprocess

import pytest

import cli_cases as CC
import common as T
import oracle_lib as OL
from bowtie_amd import ebwt_build as EB
from bowtie_amd import hostio as H
from test_ebwt_build import read_fa

D = os.path.join(T.G, "simple")
BIN = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")


def manifest():
    with open(os.path.join(D, "MANIFEST.json")) as f:
        return json.load(f)


def all_runs():
    """(case, run): every case as the harness runs it ("asis": pairs go through PairedBWAlignerV1, the reference's
    default; --12 / --interleaved input puts the reference on its stateful aligners whether the records are pairs or
    not, ebwt_search.cpp:3001-3002), and the paired / one-file cases once more with --best added (PairedBWAlignerV2)."""
    out = []
    for c in manifest()["cases"]:
        for r in c.get("runs", []):
            if r["variant"] == "asis" or (r["variant"] == "best" and (c["paired"] or c.get("needs_best"))):
                out.append((c, r))
    return out


def engine_policy(case, run, pol):
    """what the reference's choice of aligner means for bt_policy: pairs without --best -> PairedBWAlignerV1; unpaired
    records from a --12 file -> the stateful unpaired aligner, i.e. what --best selects"""
    one_file = case["reads"][0] in ("--12", "--interleaved")
    if case["paired"] and "--best" not in run["args"]:
        return dict(pol, pe_v1=True)
    if one_file and not case["paired"]:
        return dict(pol, best=True)
    return pol


@pytest.fixture(scope="session")
def simple_index(tmp_path_factory):
    """ref_<k>.fa -> index files, built by bowtie_amd/ebwt_build.py (byte-identical to bowtie-build on
    these references: asserted when the fixtures were generated)."""
    root = tmp_path_factory.mktemp("simple_idx")
    built = {}

    def get(ref):
        if ref not in built:
            names, seqs = read_fa(os.path.join(T.G, ref))
            base = str(root / os.path.basename(ref)[:-3])
            EB.build_index(seqs, names, base)
            built[ref] = base
        return built[ref]
    return get


def expected(run) -> bytes:
    with gzip.open(os.path.join(T.G, run["file"]), "rb") as f:
        return f.read()


_oi = {}


@pytest.mark.parametrize("case,run", all_runs(), ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_simple_case_oracle_and_host_io(case, run, simple_index):
    base = simple_index(case["ref"])
    rd, pol, out, ex = CC.interpret(run["args"])
    spec = lambda x: x if rd.get("fmt") == "cmdline" else ",".join(os.path.join(T.G, f) for f in x.split(","))
    want = expected(run)
    one_file = case["reads"][0] in ("--12", "--interleaved")
    if one_file:
        rd.update(dict(fmt="tabbed") if case["reads"][0] == "--12" else dict(interleaved=True))
    try:
        if one_file and case["paired"]:
            b1 = H.read_all(spec(case["reads"][1]), mate=1, **rd)
            b2 = H.read_all(spec(case["reads"][1]), mate=2, **rd)
            assert b1 is None or rd.get("interleaved") or b1.n_paired == b1.n
        elif one_file:
            b1 = H.read_all(spec(case["reads"][1]), **rd)
            assert b1 is None or b1.n_paired == 0
        elif case["paired"]:
            b1 = H.read_all(spec(case["reads"][1]), mate=1, **rd)
            b2 = H.read_all(spec(case["reads"][3]), mate=2, **rd)
        else:
            b1 = H.read_all(spec(case["reads"][0]), **rd)
    except H.ReadInputError:
        assert run["returncode"] != 0
        return
    assert run["returncode"] == 0, "the reference aborted on this input; the parser must too"
    if b1 is None:
        assert want == b""
        return
    if base not in _oi:
        _oi[base] = OL.OracleIndex(base)
    oi = _oi[base]
    pol = engine_policy(case, run, pol)
    opol = OL.make_policy(**pol)
    import refrun as R
    opts = H.out_opts(**out)
    if case["paired"]:
        cap = 4096 if pol.get("all_hits") else 2 * pol.get("khits", 1)
        per = R.oracle_search_pairs(oi, opol, b1, b2, cap=cap, v1=bool(pol.get("pe_v1")))
        hits, nh, st, pool = H.pack_hits(per, cap)
        text, _ = H.format_pairs(b1, b2, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    else:
        cap = 4096 if pol.get("all_hits") else pol.get("khits", 1)
        per = R.oracle_search(oi, opol, b1, cap=cap)
        hits, nh, st, pool = H.pack_hits(per, cap)
        text, _ = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    assert text == want


@pytest.mark.parametrize("case,run", [(c, r) for c, r in all_runs() if "sam" not in r["file"]],
                         ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_binary_reads_the_input_the_way_the_parsers_do(case, run):
    """bowtie-amd's own reading of its command line and input (which files, which format, pairs or not -- for --12 it
    looks at the first record), checked without a GPU: BT_CLI_INPUT_ONLY=1 lists the reads it would search."""
    env = dict(os.environ, BT_CLI_INPUT_ONLY="1")
    cmd = [BIN, "--wrapper", "basic-0", "-p", "1"] + run["args"] + ["-x", "/nonexistent"] + case["reads"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, env=env, timeout=60)
    rd, pol, out, ex = CC.interpret(run["args"])
    spec = lambda x: x if rd.get("fmt") == "cmdline" else ",".join(os.path.join(T.G, f) for f in x.split(","))
    one_file = case["reads"][0] in ("--12", "--interleaved")
    if one_file:
        rd.update(dict(fmt="tabbed") if case["reads"][0] == "--12" else dict(interleaved=True))
    f1, f2 = (case["reads"][1], case["reads"][1]) if one_file else (case["reads"][1], case["reads"][3]) if case["paired"] else (case["reads"][0], None)
    try:
        b1 = H.read_all(spec(f1), mate=1 if case["paired"] else 0, **rd)
        b2 = H.read_all(spec(f2), mate=2, **rd) if case["paired"] else None
    except H.ReadInputError:
        assert p.returncode == 1
        return
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-300:]
    rows = [l.split("\t") for l in p.stdout.decode().splitlines()]
    want = []
    for i in range(b1.n if b1 is not None else 0):
        row = [b1.names[i].decode(), str(int(b1.len[i]))]
        if case["paired"]:
            row += [b2.names[i].decode(), str(int(b2.len[i]))]
        want.append(row)
    assert rows == want


_emu = {}


@pytest.mark.parametrize("case,run", all_runs(), ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_simple_case_automaton_emu(case, run, simple_index):
    """The same cases with the host build of the device automatons (tests/emu) doing the search instead of the
    oracle: C++ parser -> bt_core.h / bt_best.h on the host -> C++ formatter == the reference's output."""
    import emu_lib as E
    from bowtie_amd import _abi as A
    if run["returncode"] != 0:
        pytest.skip("an input error: nothing reaches the search")
    base = simple_index(case["ref"])
    rd, pol, out, ex = CC.interpret(run["args"])
    spec = lambda x: x if rd.get("fmt") == "cmdline" else ",".join(os.path.join(T.G, f) for f in x.split(","))
    one_file = case["reads"][0] in ("--12", "--interleaved")
    if one_file:
        rd.update(dict(fmt="tabbed") if case["reads"][0] == "--12" else dict(interleaved=True))
    f1, f2 = (case["reads"][1], case["reads"][1]) if one_file else (case["reads"][1], case["reads"][3]) if case["paired"] else (case["reads"][0], None)
    if case["paired"]:
        b1, b2 = H.read_all(spec(f1), mate=1, **rd), H.read_all(spec(f2), mate=2, **rd)
    else:
        b1 = H.read_all(spec(f1), **rd)
    want = expected(run)
    if b1 is None:
        assert want == b""
        return
    if base not in _emu:
        _emu[base] = E.EmuAligner(base)
    if base not in _oi:
        _oi[base] = OL.OracleIndex(base)
    oi = _oi[base]
    p = A.make_policy(**engine_policy(case, run, pol))
    opts = H.out_opts(**out)
    if case["paired"]:
        cap = 4096 if pol.get("all_hits") else 2 * pol.get("khits", 1)
        per = _emu[base].align_pairs(p, b1, b2, hit_cap=cap)
        hits, nh, st, pool = H.pack_hits(per, cap)
        text, _ = H.format_pairs(b1, b2, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    else:
        cap = 4096 if pol.get("all_hits") else pol.get("khits", 1)
        per = _emu[base].align(p, b1, hit_cap=cap, lite=not p.best)
        hits, nh, st, pool = H.pack_hits(per, cap)
        text, _ = H.format_hits(b1, hits, nh, st, pool, cap, oi.refnames, oi.reflens, opts)
    assert text == want


@pytest.mark.gpu
@pytest.mark.parametrize("case,run", all_runs(), ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_simple_case_bowtie_amd(case, run, simple_index):
    base = simple_index(case["ref"])
    # BT_TEST_CLI_EXTRA: extra bowtie-amd options for every case (e.g. "--stream" to put the suite through the
    # experimental streamed search)
    cmd = [BIN, "--wrapper", "basic-0", "-p", "1"] + os.environ.get("BT_TEST_CLI_EXTRA", "").split() + run["args"] + ["-x", base] + case["reads"]
    # BT_LOCUS=1: the binary builds the locus image whatever the input's size (by itself it would not for a handful of reads),
    # so that these cases go through locus mode as the library-level GPU tests do
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=600, env=dict(os.environ, BT_LOCUS=os.environ.get("BT_LOCUS", "1")))
    assert (p.returncode != 0) == (run["returncode"] != 0), p.stderr.decode(errors="replace")
    if run["returncode"] == 0:
        assert p.stdout == expected(run)


DOLLAR_ROW = [(c, r) for c, r in all_runs() if c["id"] in (5, 100)]


@pytest.mark.gpu
@pytest.mark.parametrize("case,run", DOLLAR_ROW, ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_ext_kernel_instances_on_the_two_tiny_inputs(case, run, simple_index, monkeypatch):
    """Regression test: the EXT=true instances of bt_search_kernel (carry-over, on-stream second pass)
    used to fault on these two inputs -- the ones that extend a one-row range at the '$' row with the base the '$' is
    stored as.  BT_FORCE_EXT=1 makes the plain path launch the EXT instances."""
    monkeypatch.setenv("BT_FORCE_EXT", "1")
    base = simple_index(case["ref"])
    cmd = [BIN, "--wrapper", "basic-0", "-p", "1"] + run["args"] + ["-x", base] + case["reads"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=120)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-300:]
    assert p.stdout == expected(run)


INSTANCES = [dict(BT_OCC=str(o), BT_NO_RL="1") for o in (1, 2, 3, 4)] + [dict(BT_OCC=str(o), BT_NO_RL3="1") for o in (1, 2)] + [dict()]


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["0", "1"], ids=["plain", "EXT"])
@pytest.mark.parametrize("inst", INSTANCES, ids=lambda d: "_".join("%s%s" % (k[3:], v) for k, v in d.items()) or "RL3")
def test_every_kernel_instance_on_the_dollar_row_inputs(inst, ext, simple_index):
    """Fence around DESIGN.md 4.4's miscompile: the two inputs that extend a one-row range at the '$' row through EVERY
    template instance of bt_search_kernel -- register-window builds for 1..4 waves per SIMD, the two-block read-in-LDS
    builds, the three-block one, each plain and EXT (carry-over / second-pass code compiled in) -- not only the ones
    the defaults happen to launch."""
    for case, run in DOLLAR_ROW:
        if "sam" in run["file"]:
            continue
        base = simple_index(case["ref"])
        env = dict(os.environ, **inst)
        if ext == "1":
            env["BT_FORCE_EXT"] = "1"
        cmd = [BIN, "--wrapper", "basic-0", "-p", "1", "--no-stream"] + run["args"] + ["-x", base] + case["reads"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, env=env, timeout=120)
        assert p.returncode == 0, (inst, ext, p.stderr.decode(errors="replace")[-300:])
        assert p.stdout == expected(run), (inst, ext, run["file"])


@pytest.mark.gpu
@pytest.mark.parametrize("case,run", DOLLAR_ROW, ids=lambda x: (x["name"].replace(" ", "_") + "#%d" % x["id"]) if "name" in x else x["file"][14:-7])
def test_streamed_binary_on_the_two_tiny_inputs(case, run, simple_index):
    """The same two inputs through `bowtie-amd --stream` (carry-over: park, adopt, closing launch)."""
    base = simple_index(case["ref"])
    cmd = [BIN, "--wrapper", "basic-0", "-p", "1", "--stream"] + run["args"] + ["-x", base] + case["reads"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=120)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-300:]
    assert p.stdout == expected(run)
