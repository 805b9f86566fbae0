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
    (["--best", "-1", "a.fq,c.fq", "-2", "b.fq", "-x", "e_coli"], "must be specified with -1 and -2"),
    (["-c", "-1", "ACGTACGTAC,TTTTACGTAC", "-2", "ACGTACGTAC", "-x", "e_coli"], "must be specified with -1 and -2"),
    (["--best", "--12", "a.tab", "-1", "a.fq", "-2", "b.fq", "-x", "e_coli"], "cannot be combined"),
    (["-Q", "a.qual", "-x", "e_coli", "cli/io.fq"], "go with -f"),
    (["--pev2", "-x", "e_coli", "cli/io.fq"], "does not have"),
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


def _list_input(*args):
    import os
    env = dict(os.environ, BT_CLI_INPUT_ONLY="1")
    return subprocess.run([BIN, "--wrapper", "basic-0"] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=T.G, env=env, timeout=60)


def test_read_files_that_do_not_open(tmp_path):
    """CFilePatternSource::open (pat.cpp:296-357): a file that does not open is skipped with a warning -- unless none of
    the files after it opens either, then the run fails (checked on the reference: exit status 1)."""
    ok = tmp_path / "ok.fa"
    ok.write_text(">a\nACGTACGTACGTACGTACGTAAA\n")
    p = _list_input("-f", "-x", "none", "/nonexistent.fa," + str(ok))
    assert p.returncode == 0 and p.stdout == b"a\t23\n" and b"Could not open read file \"/nonexistent.fa\"" in p.stderr
    p = _list_input("-f", "-x", "none", str(ok) + ",/nonexistent.fa")
    assert p.returncode == 1 and p.stdout == b"a\t23\n"
    p = _list_input("-x", "none", "/nonexistent.fq")
    assert p.returncode == 1 and p.stdout == b""


def test_quality_files_are_only_opened(tmp_path):
    """-Q / --Q1 / --Q2 with -f (pat.cpp:333-347): the reference opens the quality file next to its read file and never
    reads it; one that does not open takes its read file out of the run."""
    ok = tmp_path / "ok.fa"
    ok.write_text(">a\nACGTACGTACGTACGTACGTAAA\n")
    q = tmp_path / "ok.qual"
    q.write_text(">a\n40 40 40\n")
    p = _list_input("-f", "-Q", str(q), "-x", "none", str(ok))
    assert p.returncode == 0 and p.stdout == b"a\t23\n" and p.stderr == b""
    p = _list_input("-f", "--quals", "/nonexistent.qual," + str(q), "-x", "none", "cli/io.fa," + str(ok))
    assert p.returncode == 0 and p.stdout == b"a\t23\n"
    assert p.stderr.decode().strip() == 'Warning: Could not open quality file "/nonexistent.qual" for reading; skipping...'
    p = _list_input("-f", "-Q", "/nonexistent.qual", "-x", "none", str(ok))
    assert p.returncode == 1 and p.stdout == b""
