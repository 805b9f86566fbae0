"""libbowtie_amd_l.so / bowtie-amd-l on the GPU: the build with 64-bit BWT rows (the reference's bowtie-align-l),
through the C ABI and through the binary, rows numbered on either side of 2^32 (tests/test_wide_rows_emu.py explains the bias).
Runs last (the name): this library was written after the round's measurements and has had 88 GPU-seconds (profiles/r5/
wide_rows_gpu.txt: the checks below, run from tests/wide_gpu_check.py and scripts/r5/wide_gpu_smoke.sh; the sweep over the plain
goldens did not finish inside them) -- the CPU suite runs the same sources through the host build -- and a failure here must
not hide the rest of the suite under -x."""
import hashlib
import os
import subprocess
import sys

import pytest

import common as T

pytestmark = pytest.mark.gpu

SEG_SHIFT = 4


def _bias(name):
    half = int(T.oracle_index(name).fw.len) // 2
    g = 1 << (SEG_SHIFT + 6)
    return (1 << 32) - (half // g) * g


def _env(bias_of=None):
    env = dict(os.environ, BT_LIB="libbowtie_amd_l.so")
    env.pop("BT_WIDE_ROW_BIAS", None)
    env.pop("BT_WIDE_SEG_SHIFT", None)
    if bias_of:
        env["BT_WIDE_ROW_BIAS"] = str(_bias(bias_of))
        env["BT_WIDE_SEG_SHIFT"] = str(SEG_SHIFT)
    return env


def _run(args, env):
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "tests", "wide_gpu_check.py")] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert p.returncode == 0 and b": ok, " in p.stdout, p.stdout.decode()[-4000:]


@pytest.mark.parametrize("biased", [False, True], ids=["plain", "biased"])
@pytest.mark.parametrize("name", ["e_coli", "multi"])
def test_gpu_wide_rank_vs_oracle(name, biased):
    _run(["rank", name], _env(name if biased else None))


@pytest.mark.parametrize("biased", [False, True], ids=["plain", "biased"])
def test_gpu_wide_matches_reference_sam(biased):
    """the golden SAMs of the phase-program modes, e_coli and multi (the bias is multi's: any multiple of the segment size
    does for e_coli too)"""
    _run(["golden"], _env("multi" if biased else None))


@pytest.mark.parametrize("biased", [False, True], ids=["plain", "biased"])
def test_gpu_wide_on_large_index_matches_bowtie_align_l(biased):
    _run(["family"], _env("multi" if biased else None))


def test_gpu_wide_vs_oracle_ragged_with_op_counts():
    _run(["ragged"], _env("multi"))


def test_gpu_wide_second_pass():
    _run(["second_pass"], _env("e_coli"))


def test_gpu_wide_best_first_and_pairs():
    """(written after the last GPU second was spent: the host build of the same code passes tests/test_wide_rows_emu.py)"""
    _run(["best"], _env("multi"))


def test_cli_l_on_large_index_is_byte_identical_to_bowtie_align_l(tmp_path):
    """bowtie-amd-l -x multi_l (the streamed path, carry-over, ticks): the SAM of the binary, end to end, rows biased"""
    import test_index_family as FAM
    from bowtie_amd.synth import write_fastq
    binp = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd-l")
    fq = str(tmp_path / "r.fq")
    env = _env("multi")
    env.pop("BT_LIB")
    for run in [r for r in FAM.fam()["runs"] if (r["reads"], r["mode"]) in (("syn36", "n2"), ("syn100", "v2"), ("syn50lowq", "n3_y"))]:
        write_fastq(T.read_set("multi", run["reads"]), fq)
        p = subprocess.run([binp, "-S", "--sam-nohead"] + run["args"] + ["-x", FAM.LARGE, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()
        assert hashlib.md5(p.stdout).hexdigest() == run["md5"], run["file"]


def test_cli_l_small_batches_carry_over(tmp_path):
    """the same through batches of 64 reads: lanes parked at the end of a launch and adopted by the next (the wide build's pool
    record is 20 pieces), ticks at the end of the input -- not among the checks of the round's last GPU seconds"""
    import test_index_family as FAM
    from bowtie_amd.synth import write_fastq
    binp = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd-l")
    fq = str(tmp_path / "r.fq")
    env = _env("multi")
    env.pop("BT_LIB")
    for run in [r for r in FAM.fam()["runs"] if (r["reads"], r["mode"]) in (("syn100", "v2"), ("syn150", "n2"))]:
        write_fastq(T.read_set("multi", run["reads"]), fq)
        p = subprocess.run([binp, "-S", "--sam-nohead", "--batch", "64"] + run["args"] + ["-x", FAM.LARGE, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()
        assert hashlib.md5(p.stdout).hexdigest() == run["md5"], run["file"]


def test_cli_l_best_first_and_pairs_are_byte_identical_to_bowtie_align_l(tmp_path):
    """(the wide best-first engine: written after the last GPU second was spent; tests/test_wide_rows_emu.py runs this test
    through the CPU shim)"""
    import test_index_family as FAM
    from bowtie_amd.synth import write_fastq
    binp = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd-l")
    fq = str(tmp_path / "r.fq")
    env = _env("multi")
    env.pop("BT_LIB")
    for run in [r for r in FAM.fam()["runs"] if (r["reads"], r["mode"]) in (("syn36", "n2_best"), ("syn100", "v2_a_best_strata"))]:
        write_fastq(T.read_set("multi", run["reads"]), fq)
        p = subprocess.run([binp, "-S", "--sam-nohead"] + run["args"] + ["-x", FAM.LARGE, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()
        assert hashlib.md5(p.stdout).hexdigest() == run["md5"], run["file"]
    # pairs (--best: PairedBWAlignerV2 and the window scan; the reference's 2-bit reference files .3/.4.ebwtl)
    for run in FAM.fam()["paired_runs"][:2]:
        b1, b2 = T.pair_set("multi", run["reads"])
        f1, f2 = str(tmp_path / "m1.fq"), str(tmp_path / "m2.fq")
        write_fastq(b1, f1)
        write_fastq(b2, f2)
        p = subprocess.run([binp, "-S", "--sam-nohead"] + run["args"] + ["-x", FAM.LARGE, "-1", f1, "-2", f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()
        assert hashlib.md5(p.stdout).hexdigest() == run["md5"], run["file"]


def test_bowtie_amd_starts_bowtie_amd_l_for_an_index_of_2_to_32_rows(tmp_path):
    """the first bytes of a .ebwtl index that claims 2^32 + 12345 rows: libbowtie_amd.so answers BT_ERR_ROWS64, bowtie-amd starts
    bowtie-amd-l, whose library reads on and finds the file truncated (the message is the second binary's)"""
    import struct
    base = str(tmp_path / "huge")
    for ext in ("", ".rev"):
        with open(base + ext + ".1.ebwtl", "wb") as f:
            f.write(struct.pack("<iQiiiii", 1, (1 << 32) + 12344, 7, 1, 5, 10, 0))
        with open(base + ext + ".2.ebwtl", "wb") as f:
            f.write(struct.pack("<i", 1))
    fq = str(tmp_path / "r.fq")
    with open(fq, "w") as f:
        f.write("@r0\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n")
    p = subprocess.run([os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd"), "-x", base, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0 and "could not be started" not in err and "2^32-1 rows" not in err, err
