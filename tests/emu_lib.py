"""ctypes driver of tests/emu/libbt_emu.so -- TEST INFRASTRUCTURE ONLY: the host build of the
per-read automaton (bowtie_amd/csrc/bt_core.h), used to check its logic against the oracle where
no GPU exists.  Not a product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

os.environ.setdefault("BT_TEST_KNOBS", "1")      # the loader's test knobs (row bias, segment size) are for tests: this is one

from bowtie_amd import _abi as A
from bowtie_amd.aligner import unpack_hits
from bowtie_amd.reads import ReadBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
# BT_EMU_DEFINES="-DBF_FAST_EXTEND=1": the host build with an experiment's compile-time switch on, in a library of its own
DEFS = os.environ.get("BT_EMU_DEFINES", "").split()
_SUFFIX = "" if not DEFS else "_" + "".join(ch if ch.isalnum() else "_" for ch in "".join(DEFS))
LIB_PATH = os.path.join(EMU_DIR, "libbt_emu%s.so" % _SUFFIX)
SRCS = [os.path.join(EMU_DIR, "bt_emu.cpp")] + [os.path.join(ROOT, "bowtie_amd", "csrc", f) for f in
                                                ("bt_host.cpp", "bt_host.h", "bt_core.h", "bt_rank.h", "bt_best.h")]
_lib = None


def build(path=None, extra=()):
    path = path or LIB_PATH
    tmp = path + ".tmp%d" % os.getpid()
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-pthread"] + DEFS + list(extra) + ["-o", tmp, SRCS[0], SRCS[1]])
    os.replace(tmp, path)


# the same sources with 64-bit BWT rows (-DBT_WIDE=1; bowtie_amd/csrc/bt_rank.h "the row type"): what libbowtie_amd_l.so runs
WIDE_LIB_PATH = os.path.join(EMU_DIR, "libbt_emu_l%s.so" % _SUFFIX)
_wlib = None


def _common_argtypes(L):
    L.emu_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.emu_index_load.restype = C.c_void_p
    L.emu_index_free.argtypes = [C.c_void_p]
    L.emu_rank4_64.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.emu_index_dims.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.emu_index_ref.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    L.emu_index_ref.restype = C.c_longlong
    L.emu_align_batch.argtypes = [C.c_void_p, C.POINTER(A.Policy), C.POINTER(A.ReadBatchC),
                                  C.POINTER(A.HitBatchC), C.POINTER(A.OpCounts)] + [C.c_uint32] * 5
    L.emu_align_pairs.argtypes = [C.c_void_p, C.POINTER(A.Policy), C.POINTER(A.ReadBatchC), C.POINTER(A.ReadBatchC),
                                  C.POINTER(A.HitBatchC), C.POINTER(A.OpCounts), C.c_uint32]


def wide_lib():
    global _wlib
    if _wlib is None:
        if not os.path.exists(WIDE_LIB_PATH) or any(os.path.getmtime(WIDE_LIB_PATH) < os.path.getmtime(s) for s in SRCS):
            build(WIDE_LIB_PATH, ["-DBT_WIDE=1"])
        L = C.CDLL(WIDE_LIB_PATH)
        _common_argtypes(L)
        _wlib = L
    return _wlib


SHIM_PATH = os.path.join(EMU_DIR, "libcli_shim%s.so" % _SUFFIX)
SHIM_SRCS = [os.path.join(EMU_DIR, "cli_shim.cpp"), os.path.join(EMU_DIR, "bt_emu.cpp")] + SRCS[1:]


WIDE_SHIM_PATH = os.path.join(EMU_DIR, "libcli_shim_l%s.so" % _SUFFIX)


def wide_shim():
    """the same for the bowtie-amd-l binary (64-bit rows)"""
    if not os.path.exists(WIDE_SHIM_PATH) or any(os.path.getmtime(WIDE_SHIM_PATH) < os.path.getmtime(s) for s in SHIM_SRCS):
        tmp = WIDE_SHIM_PATH + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-pthread", "-DBT_WIDE=1"] + DEFS + ["-I" + os.path.join(ROOT, "include"), "-o", tmp,
                               SHIM_SRCS[0], SHIM_SRCS[2]])
        os.replace(tmp, WIDE_SHIM_PATH)
    return WIDE_SHIM_PATH


def shim():
    """tests/emu/libcli_shim.so (the GPU-side entry points of the C ABI answered by the host build, LD_PRELOADed under
    the bowtie-amd binary by tests/test_cli_shim.py), rebuilt when stale.  Called from conftest.py in the main process, so
    that xdist workers never race on the file."""
    if not os.path.exists(SHIM_PATH) or any(os.path.getmtime(SHIM_PATH) < os.path.getmtime(s) for s in SHIM_SRCS):
        tmp = SHIM_PATH + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w"] + DEFS + ["-I" + os.path.join(ROOT, "include"), "-o", tmp,
                               SHIM_SRCS[0], SHIM_SRCS[2]])
        os.replace(tmp, SHIM_PATH)
    return SHIM_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in SRCS):
            build()
        L = C.CDLL(LIB_PATH)
        _common_argtypes(L)
        L.emu_rank4.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


class EmuAligner:
    def __init__(self, base: str, need_mirror=True, offrate=-1, wide=False, row_bias=None, seg_shift=None):
        """wide: the build with 64-bit BWT rows; row_bias / seg_shift: its test knobs (bt_host.h) -- the image's rows numbered
        from row_bias, 2^seg_shift rank blocks per segment"""
        self.L = wide_lib() if wide else lib()
        saved = {k: os.environ.get(k) for k in ("BT_WIDE_ROW_BIAS", "BT_WIDE_SEG_SHIFT")}
        try:
            for k, v in (("BT_WIDE_ROW_BIAS", row_bias), ("BT_WIDE_SEG_SHIFT", seg_shift)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            self.h = self.L.emu_index_load(base.encode(), int(need_mirror), offrate)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if not self.h:
            raise IOError("emu: cannot load " + base)

    def rank4(self, row, mirror=False):
        lf = (C.c_uint64 * 4)()
        L = C.c_uint32()
        self.L.emu_rank4_64(self.h, int(mirror), row, lf, C.byref(L))
        return list(lf), int(L.value)

    def refs(self):
        """([names], [lengths]) of the index's sequences"""
        names, lens = [], []
        buf = C.create_string_buffer(4096)
        t = 0
        while True:
            n = self.L.emu_index_ref(self.h, t, buf, 4096)
            if n < 0:
                return names, lens
            names.append(buf.value.decode())
            lens.append(int(n))
            t += 1

    def dims(self):
        """(text length, row bias, bytes per row)"""
        d = (C.c_uint64 * 3)()
        self.L.emu_index_dims(self.h, d)
        return int(d[0]), int(d[1]), int(d[2])

    def align(self, pol: A.Policy, batch: ReadBatch, hit_cap=None, mm_per_hit=8, counts=None,
              n_lanes=64, fr_cap=64, ent_cap=None, pal_cap=1024, no_rl=False, lite=False):
        n = batch.n
        hit_cap = hit_cap or (64 if pol.all_hits else max(1, min(int(pol.khits), 64)))
        ent_cap = ent_cap or (0 if pol.best else 12 * max(64, batch.stride))
        seq = np.ascontiguousarray(batch.seq, dtype=np.uint8)
        qual = np.ascontiguousarray(batch.qual, dtype=np.uint8)
        ln = np.ascontiguousarray(batch.len, dtype=np.uint16)
        seed = np.ascontiguousarray(batch.seed, dtype=np.uint32)
        hits = np.zeros(n * hit_cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint8)
        pool = np.zeros(max(1, n * hit_cap * mm_per_hit), dtype=np.uint16)
        rb = A.ReadBatchC(n, batch.stride, seq.ctypes.data, qual.ctypes.data, ln.ctypes.data, seed.ctypes.data)
        hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data,
                         pool.ctypes.data, len(pool), 0)
        rc = self.L.emu_align_batch(self.h, C.byref(pol), C.byref(rb), C.byref(hb),
                                   C.byref(counts) if counts is not None else None,
                                   n_lanes, fr_cap, ent_cap, pal_cap, 2 if lite else int(no_rl))
        if rc != 0:
            raise RuntimeError("emu_align_batch rc=%d" % rc)
        return unpack_hits(n, hit_cap, hits, n_hits, status, pool, int(pol.khits), int(pol.mhits),
                           bool(pol.all_hits), sample_max=bool(pol.sample_max))

    def align_pairs(self, pol: A.Policy, b1: ReadBatch, b2: ReadBatch, hit_cap=None, mm_per_hit=8, counts=None, arena_words=0):
        """-> per pair (hits: upstream mate, downstream mate, ..., hitsForThisRead, status)"""
        from bowtie_amd.aligner import pack_batch, pair_hit_cap, unpack_pair_hits
        n = b1.n
        hit_cap = hit_cap or pair_hit_cap(pol)
        k1, rb1 = pack_batch(b1)
        k2, rb2 = pack_batch(b2)
        hits = np.zeros(n * hit_cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint8)
        pool = np.zeros(max(1, n * hit_cap * mm_per_hit), dtype=np.uint16)
        hb = A.HitBatchC(hit_cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
        rc = self.L.emu_align_pairs(self.h, C.byref(pol), C.byref(rb1), C.byref(rb2), C.byref(hb),
                                   C.byref(counts) if counts is not None else None, arena_words)
        if rc != 0:
            raise RuntimeError("emu_align_pairs rc=%d" % rc)
        return unpack_pair_hits(n, hit_cap, hits, n_hits, status, pool, pol)


# tests/emu/gpu_stall.hip: keeps a HIP stream busy for a stated time (GPU tests; compiled here with hipcc, which cross-compiles
# gfx950 without a GPU, so that the .so travels to the GPU box with the snapshot)
STALL_PATH = os.path.join(EMU_DIR, "libgpu_stall.so")
STALL_SRC = os.path.join(EMU_DIR, "gpu_stall.hip")
_stall = None


def build_stall():
    if not os.path.exists(STALL_PATH) or os.path.getmtime(STALL_PATH) < os.path.getmtime(STALL_SRC):
        tmp = STALL_PATH + ".tmp%d" % os.getpid()
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", tmp, STALL_SRC])
        os.replace(tmp, STALL_PATH)
    return STALL_PATH


def stall_lib():
    global _stall
    if _stall is None:
        L = C.CDLL(build_stall())
        L.gpu_stall.argtypes = [C.c_void_p, C.c_uint]
        L.gpu_stall_probe.argtypes = [C.c_int, C.c_uint, C.c_size_t]
        L.gpu_stall_probe.restype = C.c_longlong
        _stall = L
    return _stall
