import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# the loader's test knobs (bt_host.cpp: BT_WIDE_ROW_BIAS, BT_WIDE_SEG_SHIFT) are honoured only where this says "a test"
os.environ.setdefault("BT_TEST_KNOBS", "1")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    # six workers when pytest-xdist is there and the command line does not say otherwise (-n 0 for none): the CPU suite in
    # about three minutes, the GPU suite (one MI355X shared by the workers, each test with its own contexts) well inside the
    # driver's limit.  Without the plugin the suite simply runs in one process.
    # An xdist WORKER runs this hook too, with numprocesses reset to None: it must never ask for workers of its own (it would
    # become a controller, and so would each of its six: a fork storm).
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return
    if config.pluginmanager.hasplugin("xdist") and getattr(config.option, "numprocesses", None) is None \
            and not os.environ.get("BT_TEST_WORKERS") == "0":
        config.option.numprocesses = int(os.environ.get("BT_TEST_WORKERS", "6"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not hasattr(config, "workerinput"):
        # build the checkers once, before any xdist worker can race on the .so files
        import emu_lib
        import oracle_lib
        oracle_lib.lib()
        emu_lib.lib()
        emu_lib.shim()
        emu_lib.wide_lib()          # the same with 64-bit rows (libbowtie_amd_l.so's sources)
        emu_lib.wide_shim()
        emu_lib.build_stall()       # tests/emu/gpu_stall.hip (GPU tests: a stream held busy)


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # BT_TEST_CLI_SHIM=1 (with LD_PRELOAD=tests/emu/libcli_shim.so): the tests that only run the bowtie-amd binary can be
    # run without a GPU, the binary's host logic over the emulator's searches -- see tests/emu/cli_shim.cpp
    if _has_gpu() or os.environ.get("BT_TEST_CLI_SHIM") == "1":
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
