"""ctypes mirror of include/bowtie_amd.h (plain-data structs only)."""
from __future__ import annotations

import ctypes as C

BT_OK, BT_ERR_IO, BT_ERR_FORMAT, BT_ERR_ARG, BT_ERR_DEVICE, BT_ERR_READ_SHORT, BT_ERR_OVERFLOW, BT_ERR_READS, BT_ERR_ROWS64, BT_ERR_UNSUPPORTED = range(10)
BT_FMT_FASTQ, BT_FMT_FASTA, BT_FMT_RAW, BT_FMT_CMDLINE, BT_FMT_FASTA_CONT, BT_FMT_TABBED = range(6)
BT_QUAL_PHRED33, BT_QUAL_PHRED64, BT_QUAL_SOLEXA64 = range(3)
BT_MODE_V, BT_MODE_N = 0, 1
BT_INDEX_BT2, BT_INDEX_EBWT, BT_INDEX_BT2L, BT_INDEX_EBWTL, BT_INDEX_SWAPPED = 0, 1, 2, 3, 16
BT_ST_SKIPPED, BT_ST_HITCAP, BT_ST_TOOSHORT, BT_ST_OVERFLOW, BT_ST_MMPOOL = 1, 2, 4, 8, 16


class Policy(C.Structure):
    _fields_ = [("mode", C.c_int32), ("mms", C.c_int32), ("seed_len", C.c_int32),
                ("qual_thresh", C.c_int32), ("max_bts", C.c_int32), ("nofw", C.c_int32),
                ("norc", C.c_int32), ("maq_round", C.c_int32), ("khits", C.c_uint32),
                ("mhits", C.c_uint32), ("all_hits", C.c_int32), ("best", C.c_int32),
                ("strata", C.c_int32), ("sample_max", C.c_int32), ("min_ins", C.c_int32), ("max_ins", C.c_int32),
                ("mate1_fw", C.c_int32), ("mate2_fw", C.c_int32), ("pair_tries", C.c_int32),
                ("allow_contain", C.c_int32), ("pe_v1", C.c_int32), ("reserved", C.c_int32 * 1)]


class ReadBatchC(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("stride", C.c_uint32), ("seq", C.c_void_p),
                ("qual", C.c_void_p), ("len", C.c_void_p), ("seed", C.c_void_p)]


class HitC(C.Structure):
    _fields_ = [("tidx", C.c_uint32), ("toff", C.c_uint32), ("oms", C.c_uint32), ("mm_off", C.c_uint32),
                ("cost", C.c_uint16), ("nmm", C.c_uint16), ("stratum", C.c_uint8), ("fw", C.c_uint8),
                ("pad", C.c_uint8 * 2)]


class HitBatchC(C.Structure):
    _fields_ = [("hit_cap", C.c_uint32), ("hits", C.c_void_p), ("n_hits", C.c_void_p),
                ("status", C.c_void_p), ("mm_pool", C.c_void_p), ("mm_pool_cap", C.c_uint32),
                ("mm_pool_used", C.c_uint32)]


class OpCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("lfex", "lf2", "lf1", "chase", "ftab", "offs", "rstarts", "frames", "lane_iters", "same_pair", "rescans", "cand_scans", "wave_rounds", "fetches",
                 "loc_lfex", "loc_lf1", "loc_chase", "loc_records", "loc_windows")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class IndexInfo(C.Structure):
    _fields_ = [("len", C.c_uint32), ("n_pat", C.c_uint32), ("n_frag", C.c_uint32),
                ("ftab_chars", C.c_uint32), ("off_rate", C.c_uint32), ("z_off", C.c_uint32),
                ("ebwt_bytes", C.c_uint64), ("offs_bytes", C.c_uint64), ("has_mirror", C.c_int32),
                ("variant", C.c_int32)]


class ReadOpts(C.Structure):
    _fields_ = [("format", C.c_int32), ("trim5", C.c_int32), ("trim3", C.c_int32), ("qual_enc", C.c_int32),
                ("seed", C.c_uint32), ("flags", C.c_uint32), ("skip", C.c_uint64), ("upto", C.c_uint64),
                ("cont_len", C.c_uint32), ("cont_freq", C.c_uint32)]


class OutOpts(C.Structure):
    _fields_ = [("sam", C.c_int32), ("full_ref", C.c_int32), ("ref_idx", C.c_int32), ("off_base", C.c_int32),
                ("print_cost", C.c_int32), ("show_seed", C.c_int32), ("mapq", C.c_int32),
                ("no_qname_trunc", C.c_int32), ("no_unal", C.c_int32), ("sam_nosq", C.c_int32),
                ("khits", C.c_uint32), ("mhits", C.c_uint32), ("all_hits", C.c_int32), ("sample_max", C.c_int32),
                ("suppress", C.c_uint64)]


class OutTally(C.Structure):
    _fields_ = [("aligned", C.c_uint64), ("unaligned", C.c_uint64), ("maxed", C.c_uint64), ("reported", C.c_uint64),
                ("sample_max", C.c_uint64), ("reported_paired", C.c_uint64)]


HIT_DTYPE = [("tidx", "<u4"), ("toff", "<u4"), ("oms", "<u4"), ("mm_off", "<u4"), ("cost", "<u2"),
             ("nmm", "<u2"), ("stratum", "u1"), ("fw", "u1"), ("pad", "u1", (2,))]


def make_policy(mode="n", mms=2, seed_len=28, qual_thresh=70, max_bts=None, nofw=False, norc=False,
                maq_round=True, khits=1, mhits=0xFFFFFFFF, all_hits=False, best=False, strata=False,
                sample_max=False, min_ins=0, max_ins=250, mate1_fw=True, mate2_fw=False, pair_tries=100,
                allow_contain=False, pe_v1=False) -> Policy:
    """Reference defaults: -n 2 -l 28 -e 70 -k 1, --maxbts 125 (800 for the best-first workers)
    (ebwt_search.cpp:153-253).  --strata, -M and -v 3 imply the best-first workers
    (ebwt_search.cpp:851-853, 877-887)."""
    # pe_v1: paired-end without --best (PairedBWAlignerV1) -- still the stateful engine, with its 800 backtracks
    best = bool(best or strata or sample_max or (mode == "v" and mms == 3) or pe_v1)
    if max_bts is None:
        max_bts = 800 if best else 125
    return Policy(BT_MODE_V if mode == "v" else BT_MODE_N, mms, seed_len, qual_thresh, max_bts,
                  int(nofw), int(norc), int(maq_round), khits, mhits, int(all_hits), int(best),
                  int(strata), int(sample_max), min_ins, max_ins, int(mate1_fw), int(mate2_fw), pair_tries,
                  int(allow_contain), int(pe_v1))
