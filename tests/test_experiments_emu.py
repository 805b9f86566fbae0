"""The host-only assertions of the best-first engine's shortcuts (bt_best.h, -DBF_CHECK=1: the extended branch is the queue's
front, the queue's cost word is its front's cost, a leaf's query is set on a fresh record), compiled into a host build of
its own (emu_lib.py: BT_EMU_DEFINES) and run under the tests that drive the engine through it, in a child pytest.  Until
round 4 this file also kept the compile-time forks of the engines bit-identical in the host build; the GPU has since
decided them (profiles/r4/) and the losing halves are gone."""
import os
import subprocess
import sys

import pytest

import common as T


@pytest.mark.parametrize("defines, files", [
    ("-DBF_CHECK=1", ["tests/test_automaton_emu.py", "tests/test_engine_fuzz.py", "-k", "best or paired or v3 or M3 or strata"]),
], ids=["shortcut_assumptions_hold"])
def test_engine_assertions_hold_in_the_host_build(defines, files):
    env = dict(os.environ, BT_EMU_DEFINES=defines)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=T.ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
