"""bowtie-amd's command line handling that needs no GPU: usage, version, options of the reference
that this build does not have, malformed values -- all decided before the index is touched."""
import os
import subprocess

import pytest

import common as T

BIN = os.path.join(T.ROOT, "bowtie_amd", "bowtie-amd")


def run(*args):
    return subprocess.run([BIN] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, timeout=60)


def test_binary_is_built():
    assert os.path.exists(BIN), "python -c 'import __graft_entry__ as g; g.build()' builds it"


def test_version_and_help():
    p = run("--version")
    assert p.returncode == 0 and b"1.3.1" in p.stdout
    p = run("--help")
    assert p.returncode == 0 and b"Usage:" in p.stdout and b"-n/--seedmms" in p.stdout
    p = run("-h")
    assert p.returncode == 0 and b"Usage:" in p.stdout


@pytest.mark.parametrize("args,msg", [
    (["--strata", "-x", "e_coli", "cli/io.fq"], "--strata must be combined with --best"),
    (["--best", "--strata", "-x", "e_coli", "cli/io.fq"], "--strata has no effect unless combined with"),
    (["-1", "a.fq", "-2", "b.fq", "-x", "e_coli"], "add --best"),
    (["--best", "-1", "a.fq,c.fq", "-2", "b.fq", "-x", "e_coli"], "must be specified with -1 and -2"),
    (["--best", "-M", "3", "-1", "a.fq", "-2", "b.fq", "-x", "e_coli"], "-M with paired-end"),
    (["--12", "a.tab", "-x", "e_coli"], "does not have"),
    (["--integer-quals", "-f", "-x", "e_coli", "cli/io.fa"], "is for FASTQ input"),
    (["-C", "-x", "e_coli", "cli/io.fq"], "colorspace"),
    (["-k", "0", "-x", "e_coli", "cli/io.fq"], "-k arg must be at least 1"),
    (["-n", "4", "-x", "e_coli", "cli/io.fq"], "at most 3"),
    (["-l", "3", "-x", "e_coli", "cli/io.fq"], "at least 5"),
    (["--no-such-option", "-x", "e_coli", "cli/io.fq"], "unrecognized option"),
    (["-x", "e_coli"], "No query or output file specified"),
    ([], "No index, query, or output file specified"),
    (["-x", "e_coli", "a", "b", "c"], "Extra parameter"),
    (["--suppress", "0", "-x", "e_coli", "cli/io.fq"], "bad --suppress"),
    (["--device", "x", "-x", "e_coli", "cli/io.fq"], "bad --device"),
])
def test_rejected_command_lines(args, msg):
    p = run(*args)
    assert p.returncode == 1
    assert msg in p.stderr.decode(errors="replace")


def test_last_of_v_and_n_wins():
    # -v 3 followed by -n 2 is a -n 2 run (and then fails for want of a GPU or index, not for its options)
    p = run("-v", "3", "-n", "2", "-x", "no_such_index", "cli/io.fq")
    assert p.returncode == 1 and b"Could not locate" in p.stderr or b"HIP" in p.stderr or b"could not load" in p.stderr
