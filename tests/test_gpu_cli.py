"""The bowtie-amd binary (bowtie_amd/csrc/bt_cli.cpp: C++ host over the C ABI, search on the GPU) end
to end against the unmodified reference's stdout/stderr on the host-I/O cases of tests/golden/cli:
same command lines, byte-identical output."""
import os
import subprocess

import pytest

import cli_cases as CC
import common as T

pytestmark = pytest.mark.gpu

BIN = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")


def run(args, index, reads, extra=()):
    # BT_TEST_CLI_EXTRA: extra bowtie-amd options for every run (e.g. "--stream"); paired runs ignore --stream by themselves
    cmd = [BIN, "--wrapper", "basic-0", "-p", "1"] + os.environ.get("BT_TEST_CLI_EXTRA", "").split() + list(extra) + list(args) + ["-x", index] + ([reads] if reads else [])
    # half of the cases with the locus image forced (BT_LOCUS=1: by itself the binary would not build it for inputs this small),
    # half as the binary decides (row space): by the parity of the command line's length
    env = dict(os.environ)
    if "BT_LOCUS" not in env and len(" ".join(cmd)) % 2 == 0:
        env["BT_LOCUS"] = "1"
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=600, env=env)


def with_dump_paths(case, tmp_path):
    """AL / UN / MAX in a case's arguments stand for dump files: real paths for this run."""
    # pairs: the keys are AL_1, AL_2, ... -- the files bowtie makes of the name AL.txt (hit.h:629-660)
    dumps = case.get("dumps", {})
    paths = {k.split("_")[0]: str(tmp_path / (k.split("_")[0] + ".txt")) for k in dumps}
    return [paths.get(a, a) for a in case["args"]], {k: str(tmp_path / (k + ".txt")) for k in dumps}


def split_pg(text: bytes):
    lines = text.split(b"\n")
    pg = [l for l in lines if l.startswith(b"@PG")]
    return [l for l in lines if not l.startswith(b"@PG")], pg


def summary_lines(stderr: bytes):
    return [l for l in stderr.decode(errors="replace").strip().split("\n")
            if l.startswith("#") or l.startswith("Reported") or l.startswith("No align")]


def warning_lines(stderr: bytes):
    return [l for l in stderr.decode(errors="replace").strip().split("\n") if l.startswith("Warning:")]


@pytest.mark.parametrize("case", CC.cases(), ids=lambda c: c["name"])
def test_cli_matches_reference(case, tmp_path):
    assert os.path.exists(BIN), "bowtie-amd is not built (python -c 'import __graft_entry__ as g; g.build()')"
    args, dumps = with_dump_paths(case, tmp_path)
    p = run(args, case["index"], case["reads"])
    assert p.returncode == 0, p.stderr.decode(errors="replace")
    want = CC.expected(case)
    got_body, got_pg = split_pg(p.stdout)
    want_body, want_pg = split_pg(want)
    assert got_body == want_body
    assert len(got_pg) == len(want_pg)
    for l in got_pg:
        assert l.startswith(b'@PG\tID:Bowtie\tVN:1.3.1\tCL:"')
    assert summary_lines(p.stderr) == summary_lines("\n".join(case["stderr"]).encode())
    assert warning_lines(p.stderr) == warning_lines("\n".join(case["stderr"]).encode())
    for k, path in dumps.items():
        got = open(path, "rb").read() if os.path.exists(path) else b""
        assert got == CC.expected_dump(case, k), k


@pytest.mark.parametrize("name", ["fq_default", "multi_all", "multi_sam_notrunc", "fq_gz_two_files", "multi_all_m3", "dump_multi_m3", "dump_fq",
                                  "best_strata_a", "bigM_sam_dump", "bigM_k2_cost", "best_trim_short",
                                  "pe_sam_head", "pe_a_strata_cost", "pe_n1_k3_sam"])
def test_cli_small_batches_and_threads_do_not_change_output(name, tmp_path):
    """Many tiny GPU batches, several host threads: same bytes (batches concatenate in read order;
    -a reads with more hits than the first pass had slots for take the second pass)."""
    case = [c for c in CC.cases() if c["name"] == name][0]
    args, dumps = with_dump_paths(case, tmp_path)
    p = run(args, case["index"], case["reads"], extra=["--batch", "37", "-p", "3"])
    assert p.returncode == 0, p.stderr.decode(errors="replace")
    assert split_pg(p.stdout)[0] == split_pg(CC.expected(case))[0]
    assert summary_lines(p.stderr) == summary_lines("\n".join(case["stderr"]).encode())
    for k, path in dumps.items():
        got = open(path, "rb").read() if os.path.exists(path) else b""
        assert got == CC.expected_dump(case, k), k


def test_cli_output_file_and_quiet(tmp_path):
    case = [c for c in CC.cases() if c["name"] == "fq_default"][0]
    out = tmp_path / "hits.txt"
    cmd = [BIN, "--quiet"] + case["args"] + ["-x", case["index"], case["reads"], str(out)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=600)
    assert p.returncode == 0 and p.stdout == b"" and p.stderr == b""
    assert out.read_bytes() == CC.expected(case)


@pytest.mark.parametrize("args,reads,msg", [
    (["-c", "-v", "2"], "ACGTACGTACGTACGTACGT,ACG", "Error: Read (1) is less than 4 characters long"),
    (["-c", "-v", "1"], "A", "Error: Reads must be at least 2 characters long in 1-mismatch mode"),
    (["--strata"], "cli/io.fq", "--strata must be combined with --best"),
    (["--best", "--strata"], "cli/io.fq", "--strata has no effect unless combined with"),
    (["-c", "-1", "ACGTACGTAC,TTTTACGTAC", "-2", "ACGTACGTAC"], "", "2 mate files/sequences were specified with -1, but 1"),
    (["--best", "-1", "cli/pe_1.fq", "-2", "cli/pee_2.fq"], "", "fewer reads in file specified with -2"),
])
def test_cli_errors(args, reads, msg):
    p = run(args, "e_coli", reads)
    assert p.returncode == 1
    assert msg in p.stderr.decode(errors="replace")


def test_cli_missing_index():
    p = run(["-v", "0"], "no_such_index", "cli/io.fq")
    assert p.returncode == 1 and "Could not locate a Bowtie index" in p.stderr.decode()


def test_cli_two_gpus_when_the_box_has_them():
    """--device 0,1 on a box with at least two GPUs: one index replica each, batches dealt round-robin, output in input
    order (skipped on the one-GPU test box; the 8-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    case = [c for c in CC.cases() if c["name"] == "multi_all"][0]
    p = run(case["args"], case["index"], case["reads"], extra=["--device", "0,1", "--batch", "11"])
    assert p.returncode == 0, p.stderr.decode(errors="replace")
    assert p.stdout == CC.expected(case)


def test_cli_batches_dealt_to_several_devices_keep_the_order():
    """--device takes a list: one index replica per entry, batches dealt round-robin, output in input
    order.  One GPU is all a test box has, so it is listed twice."""
    case = [c for c in CC.cases() if c["name"] == "multi_all"][0]
    p = run(case["args"], case["index"], case["reads"], extra=["--device", "0,0", "--batch", "11", "--inflight", "2"])
    assert p.returncode == 0, p.stderr.decode(errors="replace")
    assert p.stdout == CC.expected(case)
