"""The C-ABI library: loads, exports every symbol include/bowtie_amd.h declares, struct layouts
match, and -- with no GPU -- fails loudly instead of falling back."""
import ctypes as C
import os
import re

import pytest

import common as T
from bowtie_amd import _abi as A
from bowtie_amd import aligner as AL

HDR = os.path.join(T.ROOT, "include", "bowtie_amd.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = AL.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libbowtie_amd.so does not export " + n
    assert set(AL.EXPORTS) == set(names)


def test_the_64_bit_row_library_exports_the_same_abi():
    """libbowtie_amd_l.so: the same sources compiled with -DBT_WIDE=1 (the reference's bowtie-align-l), the same entry points"""
    L = C.CDLL(os.path.join(os.path.dirname(AL.LIB_PATH), "libbowtie_amd_l.so"))
    for n in declared_functions():
        assert hasattr(L, n), "libbowtie_amd_l.so does not export " + n
    assert L.bt_rows64() == 1 and AL.lib().bt_rows64() == 0
    L.bt_version.restype = C.c_char_p
    assert b"64-bit rows" in L.bt_version()
    L.bt_strerror.restype = C.c_char_p
    assert b"64-bit" in L.bt_strerror(A.BT_ERR_ROWS64) and b"bt_probe_rank64" in L.bt_strerror(A.BT_ERR_UNSUPPORTED)


def test_struct_sizes():
    assert C.sizeof(A.HitC) == 24
    assert C.sizeof(A.Policy) == 88
    assert C.sizeof(A.OpCounts) == 152


def test_policy_default_matches_reference_defaults():
    p = A.Policy()
    AL.lib().bt_policy_default(C.byref(p))
    d = A.make_policy()
    for f, _ in A.Policy._fields_:
        if f != "reserved":
            assert getattr(p, f) == getattr(d, f), f
    assert (p.mode, p.mms, p.seed_len, p.qual_thresh, p.max_bts, p.khits) == (A.BT_MODE_N, 2, 28, 70, 125, 1)


def test_version_and_strerror():
    assert b"gfx950" in AL.lib().bt_version()
    assert AL.strerror(A.BT_ERR_DEVICE)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_error_not_fallback():
    with pytest.raises(AL.BowtieAmdError) as e:
        AL.Index(os.path.join(T.G, "e_coli"))
    assert e.value.code == A.BT_ERR_DEVICE


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(T.ROOT, "bowtie_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                s = open(os.path.join(dp, f)).read()
                assert "liboracle" not in s and "bt_oracle" not in s and "libbt_emu" not in s, f
