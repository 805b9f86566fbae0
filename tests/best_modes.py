"""Option sets that send the reference down its stateful best-first workers (--best, --strata, -M,
-v 3): name -> (reference command-line arguments, make_policy keywords).  Shared by
oracle/gen_golden.py (which runs the unmodified reference on them) and the parity tests."""

# run on every read set
BEST_CORE = {
    "n2_best": (["-n", "2", "--best"], dict(mode="n", mms=2, best=True)),
    "v0_best": (["-v", "0", "--best"], dict(mode="v", mms=0, best=True)),
    "v1_best": (["-v", "1", "--best"], dict(mode="v", mms=1, best=True)),
    "v2_best": (["-v", "2", "--best"], dict(mode="v", mms=2, best=True)),
    "v3": (["-v", "3"], dict(mode="v", mms=3)),
    "n0_best": (["-n", "0", "--best"], dict(mode="n", mms=0, best=True)),
    "n1_best": (["-n", "1", "--best"], dict(mode="n", mms=1, best=True)),
    "n3_best": (["-n", "3", "--best"], dict(mode="n", mms=3, best=True)),
    "v2_a_best_strata": (["-v", "2", "-a", "--best", "--strata"], dict(mode="v", mms=2, all_hits=True, strata=True)),
    "n2_k3_best": (["-n", "2", "-k", "3", "--best"], dict(mode="n", mms=2, khits=3, best=True)),
    "n2_M3": (["-n", "2", "-M", "3"], dict(mode="n", mms=2, mhits=3, sample_max=True)),
    "n2_k2_best_strata_m5": (["-n", "2", "-k", "2", "--best", "--strata", "-m", "5"],
                             dict(mode="n", mms=2, khits=2, mhits=5, strata=True)),
}
# run on the EXTRA_SETS of oracle/gen_golden.py and on reads/e_coli_1000.fq
BEST_EXTRA = {
    "v3_a_best_strata": (["-v", "3", "-a", "--best", "--strata"], dict(mode="v", mms=3, all_hits=True, strata=True)),
    "n2_best_nofw": (["-n", "2", "--best", "--nofw"], dict(mode="n", mms=2, best=True, nofw=True)),
    "n3_best_norc_k5": (["-n", "3", "--best", "--norc", "-k", "5"], dict(mode="n", mms=3, best=True, norc=True, khits=5)),
    "n2_best_nomaq": (["-n", "2", "--best", "--nomaqround"], dict(mode="n", mms=2, best=True, maq_round=False)),
    "n2_best_maxbts5": (["-n", "2", "--best", "--maxbts", "5"], dict(mode="n", mms=2, best=True, max_bts=5)),
    "n3_best_maxbts20_k4": (["-n", "3", "--best", "--maxbts", "20", "-k", "4"],
                            dict(mode="n", mms=3, best=True, max_bts=20, khits=4)),
    "n1_best_l20_e100": (["-n", "1", "--best", "-l", "20", "-e", "100"],
                         dict(mode="n", mms=1, best=True, seed_len=20, qual_thresh=100)),
    "v1_M1": (["-v", "1", "-M", "1"], dict(mode="v", mms=1, mhits=1, sample_max=True)),
    "n2_a_best_strata_m10": (["-n", "2", "-a", "--best", "--strata", "-m", "10"],
                             dict(mode="n", mms=2, all_hits=True, strata=True, mhits=10)),
    "n2_best_M2_k2": (["-n", "2", "--best", "-M", "2", "-k", "2"],
                      dict(mode="n", mms=2, best=True, mhits=2, khits=2, sample_max=True)),
    "v2_best_a": (["-v", "2", "--best", "-a"], dict(mode="v", mms=2, best=True, all_hits=True)),
    "n3_best_a_l12_e200": (["-n", "3", "--best", "-a", "-l", "12", "-e", "200"],
                           dict(mode="n", mms=3, best=True, all_hits=True, seed_len=12, qual_thresh=200)),
    "n2_best_y": (["-n", "2", "--best", "-y"], dict(mode="n", mms=2, best=True, max_bts=0x7FFFFFFF)),
    "v3_best_k10_strata": (["-v", "3", "--best", "-k", "10", "--strata"], dict(mode="v", mms=3, khits=10, strata=True)),
    "n0_best_a_m3": (["-n", "0", "--best", "-a", "-m", "3"], dict(mode="n", mms=0, best=True, all_hits=True, mhits=3)),
}
BEST_MODES = dict(BEST_CORE)
BEST_MODES.update(BEST_EXTRA)

# paired-end (PairedBWAlignerV2: --best with -1/-2): name -> (reference arguments, make_policy keywords)
PAIRED_MODES = {
    "pe_n1_best_X500": (["-n", "1", "--best", "-X", "500"], dict(mode="n", mms=1, best=True, max_ins=500)),
    "pe_n2_best_X500": (["-n", "2", "--best", "-X", "500"], dict(mode="n", mms=2, best=True, max_ins=500)),
    "pe_v0_best_X500": (["-v", "0", "--best", "-X", "500"], dict(mode="v", mms=0, best=True, max_ins=500)),
    "pe_v1_best_X500": (["-v", "1", "--best", "-X", "500"], dict(mode="v", mms=1, best=True, max_ins=500)),
    "pe_v2_best_X500": (["-v", "2", "--best", "-X", "500"], dict(mode="v", mms=2, best=True, max_ins=500)),
    "pe_v3_best_X500": (["-v", "3", "--best", "-X", "500"], dict(mode="v", mms=3, best=True, max_ins=500)),
    "pe_n0_best_X500": (["-n", "0", "--best", "-X", "500"], dict(mode="n", mms=0, best=True, max_ins=500)),
    "pe_n3_best_X500": (["-n", "3", "--best", "-X", "500"], dict(mode="n", mms=3, best=True, max_ins=500)),
    "pe_n2_best_X400_I250_k3": (["-n", "2", "--best", "-X", "400", "-I", "250", "-k", "3"],
                                dict(mode="n", mms=2, best=True, max_ins=400, min_ins=250, khits=3)),
    "pe_n1_a_strata_X500": (["-n", "1", "--best", "-X", "500", "-a", "--strata"],
                            dict(mode="n", mms=1, strata=True, all_hits=True, max_ins=500)),
    "pe_n2_best_X500_m1": (["-n", "2", "--best", "-X", "500", "-m", "1"], dict(mode="n", mms=2, best=True, max_ins=500, mhits=1)),
    "pe_n2_best": (["-n", "2", "--best"], dict(mode="n", mms=2, best=True)),
    "pe_v2_best_X500_nofw": (["-v", "2", "--best", "-X", "500", "--nofw"], dict(mode="v", mms=2, best=True, max_ins=500, nofw=True)),
    "pe_n2_best_X500_norc_k2": (["-n", "2", "--best", "-X", "500", "--norc", "-k", "2"],
                                dict(mode="n", mms=2, best=True, max_ins=500, norc=True, khits=2)),
    "pe_n2_best_X500_ff": (["-n", "2", "--best", "-X", "500", "--ff"], dict(mode="n", mms=2, best=True, max_ins=500, mate2_fw=True)),
    "pe_n2_best_X500_rf": (["-n", "2", "--best", "-X", "500", "--rf"],
                           dict(mode="n", mms=2, best=True, max_ins=500, mate1_fw=False, mate2_fw=True)),
    "pe_n2_best_X500_pairtries2_k4": (["-n", "2", "--best", "-X", "500", "--pairtries", "2", "-k", "4"],
                                      dict(mode="n", mms=2, best=True, max_ins=500, pair_tries=2, khits=4)),
    # -M for pairs: one of the buffered pairs of the best stratum, at random (hit.cpp:27-55, sam.cpp:274-299)
    "pe_n2_best_X500_M1": (["-n", "2", "--best", "-X", "500", "-M", "1"], dict(mode="n", mms=2, best=True, max_ins=500, mhits=1, sample_max=True)),
    "pe_v2_strata_X500_M2": (["-v", "2", "--best", "--strata", "-X", "500", "-M", "2"],
                             dict(mode="v", mms=2, strata=True, max_ins=500, mhits=2, sample_max=True)),
}
# (index, pair set name) -> how tests regenerate the pairs (tests/common.py: pair_set)
PAIR_SETS = [("e_coli", "e_coli_1000_pe"), ("e_coli", "pe50"), ("multi", "pe50"), ("multi", "pe100"), ("multi", "pe30"), ("e_coli", "pe75")]


# Paired-end WITHOUT --best: PairedBWAlignerV1, the reference's default paired-end aligner (aligner.h:606-1480).
# name -> (bowtie options, make_policy keywords).  The backtrack budget: the reference has two defaults,
# maxBtsBetter = 125 for the non-stateful seeded worker and maxBts = 800 for every stateful aligner
# (ebwt_search.cpp:185-186, 2416-2529 vs 2644, 2670) -- and paired-end is always stateful, --best or not.
PAIRED_V1_MODES = {
    "pev1_n2_X500": (["-n", "2", "-X", "500"], dict(mode="n", mms=2, max_ins=500)),
    "pev1_n0_X500": (["-n", "0", "-X", "500"], dict(mode="n", mms=0, max_ins=500)),
    "pev1_n1_X500": (["-n", "1", "-X", "500"], dict(mode="n", mms=1, max_ins=500)),
    "pev1_n3_X500": (["-n", "3", "-X", "500"], dict(mode="n", mms=3, max_ins=500)),
    "pev1_v0_X500": (["-v", "0", "-X", "500"], dict(mode="v", mms=0, max_ins=500)),
    "pev1_v1_X500": (["-v", "1", "-X", "500"], dict(mode="v", mms=1, max_ins=500)),
    "pev1_v2_X500": (["-v", "2", "-X", "500"], dict(mode="v", mms=2, max_ins=500)),
    "pev1_n2_default_X": (["-n", "2"], dict(mode="n", mms=2, max_ins=250)),
    "pev1_n2_X400_I250_k3": (["-n", "2", "-X", "400", "-I", "250", "-k", "3"], dict(mode="n", mms=2, max_ins=400, min_ins=250, khits=3)),
    "pev1_n2_X500_ff": (["-n", "2", "-X", "500", "--ff"], dict(mode="n", mms=2, max_ins=500, mate1_fw=True, mate2_fw=True)),
    "pev1_n2_X500_rf": (["-n", "2", "-X", "500", "--rf"], dict(mode="n", mms=2, max_ins=500, mate1_fw=False, mate2_fw=True)),
    "pev1_n2_X500_m1": (["-n", "2", "-X", "500", "-m", "1"], dict(mode="n", mms=2, max_ins=500, mhits=1)),
    "pev1_n1_X500_a": (["-n", "1", "-X", "500", "-a"], dict(mode="n", mms=1, max_ins=500, all_hits=True)),
    "pev1_v2_X500_nofw": (["-v", "2", "-X", "500", "--nofw"], dict(mode="v", mms=2, max_ins=500, nofw=True)),
    "pev1_n2_X500_norc_k2": (["-n", "2", "-X", "500", "--norc", "-k", "2"], dict(mode="n", mms=2, max_ins=500, norc=True, khits=2)),
    "pev1_v2_X500_pairtries2_k4": (["-v", "2", "-X", "500", "--pairtries", "2", "-k", "4"], dict(mode="v", mms=2, max_ins=500, pair_tries=2, khits=4)),
    "pev1_n2_X500_allow_contain": (["-n", "2", "-X", "500", "--allow-contain"], dict(mode="n", mms=2, max_ins=500, allow_contain=True)),
    "pev1_n2_l20_e100_X500": (["-n", "2", "-l", "20", "-e", "100", "-X", "500"], dict(mode="n", mms=2, seed_len=20, qual_thresh=100, max_ins=500)),
    "pev1_n2_maxbts10_X500": (["-n", "2", "--maxbts", "10", "-X", "500"], dict(mode="n", mms=2, max_bts=10, max_ins=500)),
    "pev1_n2_nomaq_X500": (["-n", "2", "--nomaqround", "-X", "500"], dict(mode="n", mms=2, maq_round=False, max_ins=500)),
}
for _k, (_a, _kw) in PAIRED_V1_MODES.items():
    _kw.setdefault("max_bts", 800)
