import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not hasattr(config, "workerinput"):
        # build the checkers once, before any xdist worker can race on the .so files
        import emu_lib
        import oracle_lib
        oracle_lib.lib()
        emu_lib.lib()
        emu_lib.shim()


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
