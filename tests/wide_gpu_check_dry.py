"""TEST INFRASTRUCTURE: a dry run of tests/wide_gpu_check.py on the CPU -- the binding's Index / Aligner replaced by the wide host
build of the device automatons (tests/emu, -DBT_WIDE=1) -- so that the script the GPU tests of libbowtie_amd_l.so run is itself
exercised where there is no GPU (tests/test_wide_rows_emu.py).  usage: as wide_gpu_check.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import emu_lib as E                                # noqa: E402
from bowtie_amd import aligner as AL               # noqa: E402

class FakeLib:
    def bt_rows64(self): return 1
    def bt_index_len64(self, h): return h.dims()[0]
    def bt_ctx_last_retried(self, h): return 1
class Idx:
    def __init__(self, base, *a, **k):
        self.e = E.EmuAligner(base, wide=True, row_bias=int(os.environ.get("BT_WIDE_ROW_BIAS", "0"), 0) or None,
                              seg_shift=int(os.environ["BT_WIDE_SEG_SHIFT"]) if "BT_WIDE_SEG_SHIFT" in os.environ else None)
        self._h = self.e
        self.refnames, self.reflens = self.e.refs()
class Al:
    def __init__(self, idx, pol): self.idx, self.pol, self._h = idx, pol, idx.e
    def align(self, batch, hit_cap=None, counts=None): return self.idx.e.align(self.pol, batch, hit_cap=hit_cap, counts=counts, pal_cap=16384)
    def align_pairs(self, b1, b2, hit_cap=None): return self.idx.e.align_pairs(self.pol, b1, b2, hit_cap=hit_cap)
    def probe_rank64(self, rows, mirror=False):
        import numpy as np
        lf = np.zeros((len(rows), 4), dtype=np.uint64); L = np.zeros(len(rows), dtype=np.uint8)
        for i, r in enumerate(rows):
            a, b = self.idx.e.rank4(int(r), mirror); lf[i] = a; L[i] = b
        return lf, L
AL.lib = lambda: FakeLib()
AL.Index = Idx
AL.Aligner = Al
sys.argv = ["wide_gpu_check.py"] + sys.argv[1:]
import runpy                                       # noqa: E402
runpy.run_path(os.path.join(HERE, "wide_gpu_check.py"), run_name="__main__")
