"""Small tools of the repository that decisions were based on: scripts/unifdef.py (collapsed the engine's compile-time forks
once the GPU had decided them) and scripts/best_wave_model.py (picked the automaton's gate defaults before the GPU A/B)."""
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
