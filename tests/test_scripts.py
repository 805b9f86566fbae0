"""Small tools of the repository that decisions were based on: scripts/unifdef.py (collapsed the engine's compile-time forks
once the GPU had decided them), scripts/best_wave_model.py (picked the automaton's gate defaults before the GPU A/B), and
round 5's two CPU measurements behind locus mode: scripts/textmode_model.py (how much of a read's search runs on one BWT
row) and scripts/pass_model.py (a wavefront's trips through the automaton per round)."""
import os
import subprocess
import sys

import common as T


def run_unifdef(text, *flags):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".h", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        p = subprocess.run([sys.executable, os.path.join(T.ROOT, "scripts", "unifdef.py")] + list(flags) + [path],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        return p.stdout.decode()
    finally:
        os.unlink(path)


SRC = """a
#ifndef FAST
#define FAST 0
#endif
#if FAST
fast
#if defined(CHECK) && !defined(__DEVICE__)
check
#endif
#else
slow
#if OTHER
other
#endif
#endif
#if FAST && !REFILL
fast_no_refill
#elif REFILL
refill
#else
neither
#endif
#ifdef UNKNOWN
kept
#else
kept_else
#endif
z"""


def test_unifdef_resolves_given_macros_and_leaves_the_rest():
    out = run_unifdef(SRC, "-DFAST=1", "-DREFILL=0")
    assert out == """a
fast
#if defined(CHECK) && !defined(__DEVICE__)
check
#endif
fast_no_refill
#ifdef UNKNOWN
kept
#else
kept_else
#endif
z"""
    out = run_unifdef(SRC, "-DFAST=0", "-DREFILL=1", "-UOTHER")
    assert out == """a
slow
refill
#ifdef UNKNOWN
kept
#else
kept_else
#endif
z"""


def test_unifdef_keeps_a_file_without_the_macros_as_it_is():
    src = open(os.path.join(T.ROOT, "bowtie_amd", "csrc", "bt_rank.h")).read()
    assert run_unifdef(src, "-DNOT_IN_THERE=1") == src


def test_wave_model_runs_the_kernels_loop_with_64_lanes():
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "scripts", "best_wave_model.py"), "--reads", "256", "--gates", "16/16/4/24,2/9/2/3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("gates")]
    assert len(lines) == 2, p.stdout.decode() + p.stderr.decode()[-800:]
    for l in lines:
        assert "modelled" in l and " hot " in l


def test_textmode_model_runs_and_locus_mode_leaves_no_one_row_rank_rounds():
    """scripts/textmode_model.py on e_coli: builds its probe copy of the emulator, prints the breakdown of a read's rounds.
    With locus mode on (the emulator's index has its locus image) next to nothing of the search is left on one-row ranges."""
    import re
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "scripts", "textmode_model.py"), "--reads", "300"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-800:]
    out = p.stdout.decode()
    m = re.search(r"mapLF1\s+([\d.]+) per read, of which on a single row\s+([\d.]+)", out)
    assert m and float(m.group(1)) < 2.0, out
    assert "fetch -> ST_LOC_REC" in out


def test_pass_model_counts_one_locus_block_per_round():
    """scripts/pass_model.py: per wavefront and round the trips round bt_lane_run's loop, the sweeps and the locus blocks, locus
    mode against row space.  bt_lane_run makes one pass per call, so a wavefront sees the locus block at most once per
    call of a lane -- twice in a round only through a lane that finishes a read and starts the next."""
    import re
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "scripts", "pass_model.py"), "--reads", "3000", "--synthetic", "0", "--lanes", "128"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-800:]
    out = p.stdout.decode()
    loc = re.search(r"locus mode: .*?([\d.]+) trips round the loop, ([\d.]+) sweeps, ([\d.]+) locus blocks .*lane rounds per read ([\d.]+)", out)
    row = re.search(r"row space: .*?([\d.]+) trips round the loop, ([\d.]+) sweeps, ([\d.]+) locus blocks .*lane rounds per read ([\d.]+)", out)
    assert loc and row, out
    assert float(loc.group(1)) <= 2.0 and float(loc.group(3)) <= 2.0 and float(row.group(3)) == 0.0, out
    assert float(loc.group(4)) < float(row.group(4)), out
