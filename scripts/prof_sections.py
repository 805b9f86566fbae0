"""Wavefront-cycle breakdown per automaton section (needs bowtie_amd/libbowtie_amd_prof.so:
`make -C bowtie_amd/csrc prof`).  Usage: BT_LIB=libbowtie_amd_prof.so python scripts/prof_sections.py <bench args>"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BT_LIB", "libbowtie_amd_prof.so")
import bench  # noqa
from bowtie_amd import aligner as AL
NAMES = ["RESUME", "SLOW", "WAIT(unused)", "FETCH+RANK", "REFILL", "LANE_RUN", "FELL_OFF", "RESOLVE_DONE", "RA_END", "FRAME_RETURN", "CHILD_RET", "RESCAN", "SEARCH_END", "PHASE_NEXT", "SEARCH_BEGIN", "FTABSEQ_DONE", "FTAB_DONE", "BT_LOOP", "CANDSCAN", "BT_PICK", "RA_BEGIN", "ROW_BEGIN", "FRAME_ENTER", "SLOW_PASSES(count)", "SINGLE_ROW_LFEX(count)", "SINGLE_ROW_RUNS(count)", "LOCUS", "LOCUS_FIRST_MM", "LOCUS_TALLY", "LOCUS_PASSES(count)", "RUN_ITERS(count)"]
orig_counts = AL.lib().bt_ctx_counts
def hook(h, cnt, reset):
    rc = orig_counts(h, cnt, reset)
    if not reset:
        out = (C.c_uint64 * len(NAMES))()
        AL.lib().bt_ctx_prof_sections(h, out, len(NAMES))
        tot = out[5] + out[3] + out[4]
        print("[prof] wavefront cycles per section (sum over waves); LANE_RUN+RANK+REFILL = %.3e" % tot, file=sys.stderr)
        for n, v in zip(NAMES, out):
            print("[prof]   %-14s %12.4e  %5.1f%%" % (n, v, 100.0 * v / max(1, tot)), file=sys.stderr)
    return rc
class L:  # proxy so bench's `lib.bt_ctx_counts` goes through the hook
    def __getattr__(self, k):
        return hook if k == "bt_ctx_counts" else getattr(AL._lib, k)
AL.lib()
real = AL._lib
AL.lib = lambda: proxy
proxy = L()
bench.main()
