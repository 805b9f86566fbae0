"""Compile-time experiments of the device engines, checked in their host build (tests/emu, no GPU): a switch that is off in
the shipped library because it has not run on a GPU yet must at least leave every answer and every operation count of the
emulator tests what it was.  The emulator is rebuilt with the switch (emu_lib.py: BT_EMU_DEFINES, a library of its own)
and the tests that drive the engine through it run again in a child pytest.  -DBF_CHECK=1 compiles in the host-only
assertions of the shortcuts' assumptions (bt_best.h)."""
import os
import subprocess
import sys

import pytest

import common as T


@pytest.mark.parametrize("defines, files", [
    ("-DBF_FAST_EXTEND=1 -DBF_CHECK=1", ["tests/test_automaton_emu.py", "tests/test_engine_fuzz.py", "-k", "best or paired or v3 or M3 or strata"]),
    # the parts that have switches of their own, all off: the sorts in place, the leaf advanced where the reference does it,
    # a read run by one call
    ("-DBF_FAST_EXTEND=1 -DBF_FAST_GATHER=0 -DBF_ONE_LEAF_SITE=0 -DBF_REFILL=0 -DBF_CHECK=1", ["tests/test_automaton_emu.py", "-k", "best or paired or v3 or M3 or strata"]),
    ("-DBT_MM_SORT_REGS=1", ["tests/test_automaton_emu.py", "-k", "not best and not paired"]),
], ids=["fast_extend", "fast_extend_sorts_and_leaf_sites_as_in_the_reference", "mismatch_lists_sorted_in_the_lane"])
def test_experiment_is_bit_identical_in_the_host_build(defines, files):
    env = dict(os.environ, BT_EMU_DEFINES=defines)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=T.ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
