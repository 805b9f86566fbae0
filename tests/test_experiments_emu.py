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


@pytest.mark.parametrize("defines, extra_env, files", [
    ("-DBF_CHECK=1", {}, ["tests/test_automaton_emu.py", "tests/test_engine_fuzz.py", "-k", "best or paired or v3 or M3 or strata"]),
    # the emulator runs the best-first engine through the kernel's loop (the wavefront automaton, 24 lanes side by side); this
    # child runs it call by call instead (bf_run_read / bf_run_pair, what bt_best_nested_kernel does): both give the reference's answers
    ("-DBF_CHECK=1", {"BT_EMU_BEST_NESTED": "1"}, ["tests/test_automaton_emu.py", "-k", "best or paired or v3 or M3 or strata"]),
    # the automaton's gates at other settings: a cold sweep for every lane that wants one, and gates that are never met early
    ("", {"BT_BEST_COLD_MIN": "1", "BT_BEST_TAKE_MIN": "1", "BT_BEST_SEND_PERIOD": "1"}, ["tests/test_automaton_emu.py", "-k", "best or paired"]),
    ("", {"BT_BEST_COLD_MIN": "24", "BT_BEST_TAKE_MIN": "24", "BT_BEST_SEND_PERIOD": "7", "BT_BEST_SEND_MIN": "24"}, ["tests/test_automaton_emu.py", "-k", "best or paired"]),
    # a sweep wanted by fewer lanes than new reads are (round 4's fifth GPU call hung on this: lanes that wait for a read
    # must not count towards a sweep that cannot give them one), and the second pass of a sweep
    ("", {"BT_BEST_COLD_MIN": "2", "BT_BEST_TAKE_MIN": "9", "BT_BEST_SEND_PERIOD": "2", "BT_BEST_SEND_MIN": "3", "BT_BEST_SWEEP_TWICE": "1"},
     ["tests/test_automaton_emu.py", "-k", "best or paired"]),
], ids=["shortcut_assumptions_hold", "call_by_call", "gates_open", "gates_late", "sweep_before_take"])
def test_engine_assertions_hold_in_the_host_build(defines, extra_env, files):
    env = dict(os.environ, BT_EMU_DEFINES=defines, **extra_env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=T.ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
