"""The bowtie-amd binary's own host logic without a GPU: tests/emu/cli_shim.cpp answers the GPU-side entry points of the C
ABI with the host build of the device automatons (LD_PRELOAD, test infrastructure only), and the tests that drive the
binary -- the reference-made command-line cases, every simple_tests.pl case, the differential fuzz against the live
reference binary -- run here as they will on the GPU.  What this checks is bt_cli.cpp (options, batching, the second pass
for reads with many hits, pairs, --12 / --interleaved, read dumps, tallies, output order) and bt_io.cpp under it; the
searches themselves are the emulator's, whose parity has its own tests.  One run puts every case through --stream: the
binary's streamed search loop (the shim answers the asynchronous entry points synchronously)."""
import os
import subprocess
import sys

import pytest

import common as T

SHIM = os.path.join(T.ROOT, "tests", "emu", "libcli_shim.so")
SRCS = [os.path.join(T.ROOT, "tests", "emu", "cli_shim.cpp"), os.path.join(T.ROOT, "tests", "emu", "bt_emu.cpp"),
        os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_host.cpp"), os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_core.h"),
        os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_best.h")]


@pytest.fixture(scope="module")
def shim():
    import emu_lib
    return emu_lib.shim()            # built by conftest.py in the main process; this only checks it is fresh


@pytest.mark.parametrize("files", [["tests/test_gpu_cli.py"], ["tests/test_simple_cases.py"], ["tests/test_zz_gpu_fuzz.py"],
                                   ["tests/test_gpu_cli.py", "--no-stream"], ["tests/test_simple_cases.py", "-k", "test_simple_case_bowtie_amd", "--no-stream"],
                                   ["tests/test_gpu_cli.py", "pinned"]],
                         ids=lambda x: x[0][6:-3] + ("_no_stream" if x[-1] == "--no-stream" else "_pinned_batches" if x[-1] == "pinned" else ""))
def test_binary_suites_through_the_cpu_shim(files, shim):
    """The binary streams unpaired default-engine batches by default (the shim answers the asynchronous entry points
    synchronously); the --no-stream runs put the same cases through the whole-batch search loop."""
    env = dict(os.environ, BT_TEST_CLI_SHIM="1", LD_PRELOAD=shim, BT_GPU_FUZZ_SEEDS="12")
    if files[-1] == "pinned":                       # BT_CLI_PINNED: the read batches come from the library's bt_host_alloc
        files = files[:-1]
        env["BT_CLI_PINNED"] = "1"
    if files[-1] == "--no-stream":
        files = files[:-1]
        env["BT_TEST_CLI_EXTRA"] = "--no-stream"
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=T.ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail


def test_formatter_pieces_reach_the_file_in_order(shim, tmp_path):
    """The writer formats a big batch in pieces on -p threads and writes them as they finish, in order: the text is the
    one thread's text.  The shim's SHIM_NULL_SEARCH answers (made-up alignments, no search) make a batch of tens of
    thousands of reads cheap here."""
    import numpy as np
    rng = np.random.default_rng(3)
    n, L = 60000, 50
    fq = tmp_path / "r.fq"
    with open(fq, "wb") as f:
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
        quals = (rng.integers(2, 41, size=(n, L)) + 33).astype(np.uint8)
        f.write(b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, bases[i].tobytes(), quals[i].tobytes()) for i in range(n)))
    env = dict(os.environ, LD_PRELOAD=shim, SHIM_NULL_SEARCH="1")
    base = os.path.join(T.ROOT, "tests", "golden", "e_coli")
    binary = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")
    for fmt in ([], ["-S"]):
        outs = []
        for threads, batch in ((1, 60000), (4, 60000), (3, 25000)):
            p = subprocess.run([binary, "-p", str(threads), "--batch", str(batch)] + fmt + ["-x", base, str(fq)], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert p.returncode == 0, p.stderr.decode(errors="replace")
            outs.append(b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")))
        assert len(outs[0]) > n * L and outs[0] == outs[1] == outs[2]
        # where the large blocks come from (bt_cli.cpp `bigmem`, the page-locked result arrays) changes no byte of the text:
        # everything from 64 KB up out of the mapped blocks, and nothing at all
        # ... nor does a device with room for four batches in flight only (bt_align_stream_room: reads then ride two launches),
        # nor taking everything down by hand at the end
        for knobs in (dict(BT_CLI_BIG_MIN="65536"), dict(BT_CLI_HUGEPAGES="0", BT_CLI_PINNED_RESULTS="0"), dict(SHIM_STREAM_ROOM="4"),
                      dict(SHIM_STREAM_ROOM="1", BT_CLI_TEARDOWN="1")):
            p = subprocess.run([binary, "-p", "3", "--batch", "9000"] + fmt + ["-x", base, str(fq)], env=dict(env, **knobs),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert p.returncode == 0, p.stderr.decode(errors="replace")
            assert b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")) == outs[0], knobs


def test_standard_input_is_spooled_and_the_spool_removed(shim, tmp_path):
    """`-` as the read file: standard input goes to a temporary file first (both mate streams of --12 open it) and the file is
    gone when the run is over -- the binary leaves through _exit(), which runs no atexit handler (round 6 left 40 of them in
    /tmp before this test).  Same output as from the file itself."""
    env = dict(os.environ, LD_PRELOAD=shim, TMPDIR=str(tmp_path))
    base = os.path.join(T.ROOT, "tests", "golden", "e_coli")
    binary = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")
    fq = os.path.join(T.ROOT, "tests", "golden", "e_coli_1000.fq")
    want = subprocess.run([binary, "-p", "2", "-x", base, fq], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert want.returncode == 0, want.stderr.decode(errors="replace")
    with open(fq, "rb") as f:
        got = subprocess.run([binary, "-p", "2", "-x", base, "-"], env=env, stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert got.returncode == 0, got.stderr.decode(errors="replace")
    assert got.stdout == want.stdout and len(want.stdout) > 1000
    assert [n for n in os.listdir(tmp_path) if n.startswith("bowtie_amd_stdin_")] == []
