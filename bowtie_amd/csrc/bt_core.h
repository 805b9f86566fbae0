/*
 * bt_core.h -- the per-read search state machine (one GPU lane = one read).
 *
 * What it computes is what a reference worker thread computes for a read between GET_READ and
 * FINISH_READ (ebwt_search.cpp:923-961) in the default unpaired modes: the phase scripts
 * (search_exact.c, search_1mm_phase*.c, search_23mm_phase*.c, search_seeded_phase*.c), the
 * randomized greedy-DFS backtracker GreedyDFSRangeSource (ebwt_search_backtrack.h:237-1091,
 * 1118-1655), the SA walk + offset resolution (Ebwt::reportChaseOne / joinedToTextOff,
 * ebwt.h:2569-2755) and the per-read hit-sink policy (hit.h:969-985, 1201-1209).
 *
 * How it computes it is not the reference's way.  The reference recurses and performs each
 * rank query inline.  Here every read is an explicit-stack automaton that runs until it needs
 * the next LF-mapping, hands a BtReq (one or two BWT rows) to its caller and is resumed with the
 * BtRes.  The caller -- the HIP kernel in bt_kernels.hip -- advances the 64 reads of a wavefront
 * in lock step, so that all rank gathers of a wavefront are issued together, independent of
 * which phase / frame / SA walk each individual read is in.
 *
 *   lane state   : BtLane (registers)
 *   frame stack  : one record per backtrack level, in HBM scratch        (BT_FR_*)
 *   range stack  : per visited query position, the 4x(top,bot) ranges and the eliminated-
 *                  alternatives mask; compact (a child frame starts where its parent stopped)
 *   seedlings    : packed partial alignments of the -n seed phases (ebwt_search_util.h:37-88)
 *
 * This header is plain C++ that compiles for gfx950 (hipcc) and for the host; the host build
 * exists only so the automaton can be unit-tested against the oracle without a GPU
 * (tests/emu).  The product library never runs it on the CPU.
 */
#ifndef BT_CORE_H_
#define BT_CORE_H_

#include "bt_rank.h"

/* ---- phase program: one BtStep per searcher invocation of a phase script ------------------ */
enum { BT_OC_ZERO = 0, BT_OC_PLEN = 1, BT_OC_S = 2, BT_OC_S3 = 3, BT_OC_S5 = 4 };
enum { BT_KIND_SEARCH = 0, BT_KIND_GEN = 1, BT_KIND_EXTEND = 2 };

struct BtStep {
	uint8_t mirror;          /* 0: index of the text, 1: mirror index                           */
	uint8_t readFw;          /* EbwtSearchParams::setFw                                         */
	uint8_t kind;            /* SEARCH | GEN (collect seedlings, setQlen(seed)) | EXTEND        */
	uint8_t reportExacts;
	uint8_t considerQuals;
	uint8_t halfAndHalf;
	uint8_t maq;
	uint8_t reportPartials;
	uint8_t oc[6];           /* setOffs(5depth,3depth,unrev,1rev,2rev,3rev) as BT_OC_* codes    */
	uint8_t pad[2];
	uint32_t qualThresh, maxBts;
};

#define BT_MAX_STEPS 12
struct BtProgram {
	int32_t  nsteps;
	int32_t  seeded;         /* -n mode: apply the phase-1 prologue filter                      */
	uint32_t seedLen;        /* -l (0xffffffff in -v modes: the "seed" is the whole read)       */
	uint32_t seedMms;
	uint32_t minLen;         /* -v: reads shorter than this are an error (BT_ST_TOOSHORT)       */
	uint32_t sinkN, sinkMax; /* NGoodHitSinkPerThread _n/_max (hit.h:937-985)                   */
	uint32_t sinkAll;
	BtStep   steps[BT_MAX_STEPS];
};

/* ---- per-lane scratch in HBM ------------------------------------------------------------- */
#define BT_FR_WORDS 24
enum {
	FR_DEPTH = 0, FR_D, FR_HAM, FR_U, FR_R1, FR_R2, FR_R3, FR_ALTNUM, FR_ELIGNUM, FR_ELIGSZ,
	FR_ELI, FR_ELTOP, FR_ELBOT, FR_ELHAM, FR_ELC, FR_LOWQ, FR_PI, FR_PJ, FR_PTOP, FR_PBOT,
	FR_EBASE, FR_MM, FR_TOP, FR_BOT
};

struct BtScratch {
	uint32_t* frames;   uint32_t frStride;   /* word w of frame f at frames[(f*24+w)*frStride]   */
	uint32_t* pairs;                         /* [entry][8]: tops ACGT, bots ACGT                 */
	uint8_t*  elims;                         /* [entry]                                           */
	uint64_t* pals;                          /* [palCap] seedlings                                */
	uint32_t  frCap, entCap, palCap;
};

/* ---- batch-level arguments --------------------------------------------------------------- */
struct BtHitRec {            /* == bt_hit (include/bowtie_amd.h) */
	uint32_t tidx, toff, oms, mm_off;
	uint16_t cost, nmm;
	uint8_t  stratum, fw, pad[2];
};

struct BtBatchDev {
	const uint8_t*  seq;  const uint8_t* qual;  const uint16_t* len;  const uint32_t* seed;
	uint32_t n_reads, stride;
	BtHitRec* hits; uint32_t hit_cap;
	uint32_t* n_hits; uint8_t* status;
	uint16_t* mm_pool; uint32_t mm_pool_cap; uint32_t* mm_pool_used;
};

#define BT_STF_SKIPPED   1u
#define BT_STF_HITCAP    2u
#define BT_STF_TOOSHORT  4u
#define BT_STF_OVERFLOW  8u     /* a per-read scratch capacity was exceeded; results invalid      */
#define BT_STF_MMPOOL    16u    /* mm_pool exhausted; hit stored without its mismatch list        */

/* ---- LF request / response --------------------------------------------------------------- */
struct BtReq { uint32_t rowA, rowB; uint32_t op; };     /* op bit0: rank at rowA, bit1: at rowB    */
struct BtRes { uint32_t a[4], b[4], LA; };

enum {
	ST_IDLE = 0, ST_PHASE_NEXT, ST_SEARCH_BEGIN, ST_FRAME_ENTER, ST_STEP_BEGIN, ST_STEP_LFDONE,
	ST_STEP_POST, ST_BT_LOOP, ST_CHILD_RET, ST_FRAME_RETURN, ST_FELL_OFF, ST_RA_BEGIN,
	ST_ROW_BEGIN, ST_CHASE_CHECK, ST_CHASE_LFDONE, ST_RESOLVE, ST_RA_END, ST_SEARCH_END
};
enum { RC_STEP = 0, RC_CHILD, RC_FELL, RC_ENTRY };
enum { LFK_EX2 = 0, LFK_C2, LFK_LF1 };

struct BtOpCnt { uint32_t lfex, lf2, lf1, chase, ftab, offs, rstarts, frames, samePair; };

struct BtLane {
	/* read */
	uint32_t rd, plen, seed;
	const uint8_t *seq, *qual;
	uint32_t S, S3, S5;
	/* sink (hit.h:969-985) */
	uint32_t nhits, stored, status;
	/* phase */
	int32_t  step;
	uint32_t npals, palIdx;
	/* searcher (GreedyDFSRangeSource members) */
	uint32_t qlen;
	uint32_t mirror, readFw, rev, reportExacts, considerQuals, halfAndHalf, maq, reportPartials, kind;
	uint32_t d5, d3, unrev, r1, r2, r3, qualThresh, maxBts;
	uint32_t rnd, numBts, bailed, nsFtab0;
	uint32_t nmuts, mutpos[3], mutnew[3];
	uint32_t iham;
	/* current frame (locals of backtrack(), ebwt_search_backtrack.h:363-455) */
	uint32_t sd, depth, d, top, bot, ham, fu, f1, f2, f3;
	uint32_t altNum, eligibleNum, eligibleSz, eli, eltop, elbot, elham, elcint, elignore, lowAltQual, ebase;
	/* per-position temporaries that live across the LF wait */
	uint32_t c, q, lfk, fl_alt, fl_elig, fl_over;
	uint32_t btDespite;
	/* pending backtrack target */
	uint32_t pi, pj, pbttop, pbtbot;
	/* report */
	uint32_t ra_sd, ra_top, ra_bot, ra_cost, ra_stratum, ra_cont, ra_r, ra_i, ra_nmm;
	uint32_t crow, cjumps;
	uint32_t ret, state;
	BtOpCnt cnt;
};

/* ---- small helpers ----------------------------------------------------------------------- */
BT_HD uint32_t bt_rnd_u32(BtLane& L)                       /* RandomSource::nextU32, random_source.h:45-54 */
{
	uint32_t ret;
	L.rnd = 1664525u * L.rnd + 1013904223u;
	ret = L.rnd >> 16;
	L.rnd = 1664525u * L.rnd + 1013904223u;
	ret ^= L.rnd;
	return ret;
}
BT_HD uint32_t bt_mm_penalty(uint32_t maq, uint32_t q)     /* qual.h:61-67, qual.cpp:4-32 */
{
	if (!maq) return q;
	if (q < 5) return 0;
	if (q < 15) return 10;
	if (q < 25) return 20;
	return 30;
}
/* query char / quality at index i of the string setQuery selected (ebwt_search_backtrack.h:90-140),
 * with the seedling mutations applied (:1368-1382) */
BT_HD uint32_t bt_qry(const BtLane& L, uint32_t i)
{
	uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	uint32_t c = L.seq[j];
	if (!L.readFw && c < 4u) c ^= 3u;
	if (L.nmuts > 0) {
		if (i == L.mutpos[0]) c = L.mutnew[0];
		if (L.nmuts > 1 && i == L.mutpos[1]) c = L.mutnew[1];
		if (L.nmuts > 2 && i == L.mutpos[2]) c = L.mutnew[2];
	}
	return c;
}
BT_HD uint32_t bt_qual(const BtLane& L, uint32_t i)
{
	uint32_t j = L.rev ? (L.plen - 1u - i) : i;
	uint32_t v = L.qual[j];
	return v >= 33u ? v - 33u : 0u;
}

#define FRW(f, w) S.frames[((f) * BT_FR_WORDS + (w)) * S.frStride]
#define PT(e, c) S.pairs[(e) * 8u + (c)]
#define PB(e, c) S.pairs[(e) * 8u + 4u + (c)]

BT_HD uint32_t bt_off_code(const BtLane& L, uint32_t code)
{
	return code == BT_OC_ZERO ? 0u : code == BT_OC_PLEN ? L.plen : code == BT_OC_S ? L.S :
	       code == BT_OC_S3 ? L.S3 : L.S5;
}

/* hhCheckTop (ebwt_search_backtrack.h:1200-1275) */
BT_HD bool bt_hh_check_top(const BtLane& L, const BtScratch& S, uint32_t d)
{
	if (d == L.d5) {
		if (L.sd == 0) return false;
	} else if (d == L.d3) {
		if (L.r3 == L.r2) {
			if (L.sd < 2) return false;
		} else {
			uint32_t lo = 0;
			for (uint32_t i = 0; i < L.sd; i++) {
				uint32_t dd = L.qlen - (FRW(i, FR_MM) & 0xffffu) - 1u;
				if (dd >= L.d5 && dd < L.d3) lo++;
			}
			if (lo == 0) return false;
		}
	}
	return true;
}

/* reportPartial (ebwt_search_backtrack.h:1571-1655) */
BT_HD void bt_report_partial(BtLane& L, const BtScratch& S, uint32_t sd)
{
	uint64_t pos[3] = {0xffff, 0xffff, 0xffff}, chr[3] = {3, 3, 3};
	for (uint32_t k = 0; k < sd && k < 3; k++) {
		uint32_t mm = FRW(k, FR_MM);
		pos[k] = mm & 0xffffu; chr[k] = (mm >> 16) & 3u;
	}
	uint64_t al = (pos[0]) | (pos[1] << 16) | (pos[2] << 32) | (chr[0] << 48) | (chr[1] << 50) | (chr[2] << 52)
	            | (0xffull << 54) | (3ull << 62);
	if (L.npals < S.palCap) S.pals[L.npals++] = al;
	else L.status |= BT_STF_OVERFLOW;
}

/* Start read `rd`: the worker-loop prologue (ebwt_search.cpp:1675-1683, 2167-2175, 2572-2584;
 * search_seeded_phase1.c:17-44). */
BT_HD void bt_lane_start(BtLane& L, const BtProgram& P, const BtBatchDev& B, uint32_t rd)
{
	L.rd = rd;
	L.plen = B.len[rd];
	L.seed = B.seed[rd];
	L.seq = B.seq + (uint64_t)rd * B.stride;
	L.qual = B.qual + (uint64_t)rd * B.stride;
	uint32_t qs = L.plen < P.seedLen ? L.plen : P.seedLen;
	L.S = qs; L.S3 = qs >> 1; L.S5 = (qs >> 1) + (qs & 1u);
	L.nhits = 0; L.stored = 0; L.status = 0;
	L.step = -1; L.npals = 0; L.palIdx = 0; L.nmuts = 0;
	L.state = ST_PHASE_NEXT;
	if (P.seeded) {
		bool skip = L.plen < 4u;
		if (!skip) {
			uint32_t ns = 0;
			for (uint32_t i = 0; i < qs; i++) if (L.seq[i] == 4u && ++ns > P.seedMms) { skip = true; break; }
		}
		if (skip) { L.status |= BT_STF_SKIPPED; L.step = P.nsteps; }
	} else if (L.plen < P.minLen) {
		L.status |= BT_STF_TOOSHORT; L.step = P.nsteps;
	}
}

/* FINISH_READ: publish the sink counters (hit.h:741-786); the hit slots were written as found. */
BT_HD void bt_lane_finish(BtLane& L, const BtBatchDev& B)
{
	B.n_hits[L.rd] = L.nhits;
	B.status[L.rd] = (uint8_t)L.status;
	L.state = ST_IDLE;
}

/* Ebwt::report + EbwtSearchParams::reportHit + sink (ebwt.h:2635-2682, 1288-1405; hit.h:969-985).
 * Returns true iff the sink says stop. */
BT_HD bool bt_report_hit(BtLane& L, const BtProgram& P, const BtIndexDev& ix, const BtScratch& S,
                         const BtBatchDev& B, uint32_t tidx, uint32_t toff)
{
	L.nhits++;
	if (L.nhits > P.sinkMax) return true;
	if (L.stored < B.hit_cap) {
		BtHitRec h;
		h.tidx = tidx; h.toff = toff; h.oms = L.ra_bot - L.ra_top - 1u;
		h.cost = (uint16_t)L.ra_cost; h.stratum = (uint8_t)L.ra_stratum; h.fw = (uint8_t)L.readFw;
		h.pad[0] = h.pad[1] = 0;
		uint32_t nmm = L.ra_nmm;
		h.nmm = (uint16_t)nmm; h.mm_off = 0;
		if (nmm > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
			uint32_t off = atomicAdd(B.mm_pool_used, nmm);
#else
			uint32_t off = *B.mm_pool_used; *B.mm_pool_used += nmm;
#endif
			if (off + nmm <= B.mm_pool_cap) {
				h.mm_off = off;
				const bool flip = (ix.fw != 0) != (L.readFw != 0);
				uint16_t* mm = B.mm_pool + off;
				for (uint32_t i = 0; i < nmm; i++) {
					uint32_t pos, refc;
					if (i < L.ra_sd) { uint32_t v = FRW(i, FR_MM); pos = v & 0xffffu; refc = (v >> 16) & 3u; }
					else { pos = L.mutpos[i - L.ra_sd]; refc = L.mutnew[i - L.ra_sd] & 3u; }
					if (flip) pos = L.qlen - pos - 1u;
					uint16_t e = (uint16_t)(pos | (refc << 12));
					/* Hit::mms is a bitset: keep the list ordered by position */
					int j = (int)i - 1;
					while (j >= 0 && (mm[j] & 0x3ffu) > (e & 0x3ffu)) { mm[j + 1] = mm[j]; j--; }
					mm[j + 1] = e;
				}
			} else {
				h.nmm = 0; L.status |= BT_STF_MMPOOL;
			}
		}
		B.hits[(uint64_t)L.rd * B.hit_cap + L.stored] = h;
		L.stored++;
	} else if (L.stored < P.sinkN) {
		L.status |= BT_STF_HITCAP;
	}
	if (P.sinkAll) return false;
	if (L.nhits == P.sinkN && (P.sinkMax == 0xffffffffu || P.sinkMax < P.sinkN)) return true;
	return false;
}

/* Begin reportAlignment (ebwt_search_backtrack.h:1455-1513) for `sd` mismatches on [top,bot). */
#define BT_GOTO_RA(SD, TOP, BOT, COST, CONT) \
	do { L.ra_sd = (SD); L.ra_top = (TOP); L.ra_bot = (BOT); L.ra_cost = (COST); L.ra_cont = (CONT); \
	     L.state = ST_RA_BEGIN; } while (0)

/*
 * Advance one lane until it needs an LF-mapping (returns with req.op != 0 and the lane in a
 * *_LFDONE state) or has finished its read (state ST_IDLE, req.op == 0).
 * `res` is consumed iff the lane was waiting for it.
 */
BT_HD void bt_lane_run(BtLane& L, const BtProgram& P, const BtIndexDev* IX, const BtScratch& S,
                       const BtBatchDev& B, const BtRes& res, BtReq& req)
{
	req.op = 0; req.rowA = 0; req.rowB = 0;
	for (;;) {
		const BtIndexDev& ix = IX[L.mirror];
		switch (L.state) {
		case ST_IDLE:
			return;

		/* ---- phase script ------------------------------------------------------------- */
		case ST_PHASE_NEXT: {
			L.step++;
			if (L.step >= P.nsteps || (L.status & BT_STF_OVERFLOW)) { bt_lane_finish(L, B); return; }
			const BtStep& st = P.steps[L.step];
			/* setQuery + setOffs + ctor flags */
			L.mirror = st.mirror; L.readFw = st.readFw; L.rev = (st.mirror != 0) == (st.readFw != 0) ? 1u : 0u;
			L.kind = st.kind; L.reportExacts = st.reportExacts; L.considerQuals = st.considerQuals;
			L.halfAndHalf = st.halfAndHalf; L.maq = st.maq; L.reportPartials = st.reportPartials;
			L.qualThresh = st.qualThresh; L.maxBts = st.maxBts;
			L.d5 = bt_off_code(L, st.oc[0]); L.d3 = bt_off_code(L, st.oc[1]); L.unrev = bt_off_code(L, st.oc[2]);
			L.r1 = bt_off_code(L, st.oc[3]); L.r2 = bt_off_code(L, st.oc[4]); L.r3 = bt_off_code(L, st.oc[5]);
			L.qlen = (st.kind == BT_KIND_GEN) ? L.S : L.plen;           /* setQlen(seed) */
			L.rnd = L.seed; L.numBts = 0; L.nmuts = 0; L.iham = 0;
			if (st.kind == BT_KIND_GEN) L.npals = 0;
			if (st.kind == BT_KIND_EXTEND) {
				L.palIdx = 0;
				if (L.npals == 0) { L.state = ST_PHASE_NEXT; break; }
				/* fallthrough into the first seedling below */
				L.state = ST_SEARCH_END; L.ret = 0; L.palIdx = 0xffffffffu;   /* "before first" */
				break;
			}
			L.state = ST_SEARCH_BEGIN;
			break;
		}

		/* ---- backtrack() entry: tallyNs + ftab jump (:237-297, 1308-1362) -------------- */
		case ST_SEARCH_BEGIN: {
			L.bailed = 0; L.sd = 0;
			/* tallyNs */
			uint32_t nsInSeed = 0; bool ok = true;
			for (uint32_t i = 0; i < L.r3 && ok; i++) {
				if (bt_qry(L, L.qlen - i - 1u) == 4u) {
					nsInSeed++;
					if (nsInSeed == 1) { if (i < L.unrev) ok = false; }
					else if (nsInSeed == 2) { if (i < L.r1) ok = false; }
					else if (nsInSeed == 3) { if (i < L.r2) ok = false; }
					else ok = false;
				}
			}
			if (!ok) { L.ret = 0; L.state = ST_SEARCH_END; break; }
			uint32_t nsInFtab = 0;
			const uint32_t ftabChars = ix.ftabChars;
			for (uint32_t i = 0; i < ftabChars && i < L.qlen; i++)
				if (bt_qry(L, L.qlen - i - 1u) == 4u) nsInFtab++;
			L.nsFtab0 = nsInFtab > 0 ? 1u : 0u;
			const uint32_t m = L.unrev < L.qlen ? L.unrev : L.qlen;
			/* frame 0 parameters */
			L.fu = L.unrev; L.f1 = L.r1; L.f2 = L.r2; L.f3 = L.r3; L.ham = L.iham; L.ebase = 0;
			if (nsInFtab == 0 && m >= ftabChars) {
				uint32_t ftabOff = bt_qry(L, L.qlen - ftabChars);
				for (uint32_t i = ftabChars - 1u; i > 0; i--) { ftabOff <<= 2; ftabOff |= bt_qry(L, L.qlen - i); }
				uint32_t top = bt_ftab_hi(ix, ftabOff), bot = bt_ftab_lo(ix, ftabOff + 1u);
				L.cnt.ftab++;
				if (L.qlen == ftabChars && bot > top) {
					if (L.reportPartials > 0) { L.depth = 0; L.top = 0; L.bot = 0; L.state = ST_FRAME_ENTER; }
					else BT_GOTO_RA(0, top, bot, L.iham, RC_ENTRY);
				} else if (bot > top) {
					L.depth = ftabChars; L.top = top; L.bot = bot; L.state = ST_FRAME_ENTER;
				} else { L.ret = 0; L.state = ST_SEARCH_END; }
			} else {
				L.depth = 0; L.top = 0; L.bot = 0; L.state = ST_FRAME_ENTER;
			}
			break;
		}

		/* ---- frame prologue (:363-455) -------------------------------------------------- */
		case ST_FRAME_ENTER: {
			L.cnt.frames++;
			if (L.halfAndHalf) {
				if (L.maxBts > 0 && L.numBts == L.maxBts) { L.bailed = 1; L.ret = 0; L.state = ST_FRAME_RETURN; break; }
				L.numBts++;
			}
			L.altNum = 0; L.eligibleNum = 0; L.eligibleSz = 0;
			L.eli = 0; L.eltop = 0; L.elbot = 0; L.elham = L.ham; L.elcint = 0; L.elignore = 1;
			L.lowAltQual = 0xff;
			L.d = L.depth;
			L.state = ST_STEP_BEGIN;
			break;
		}

		/* ---- one query position (:456-568) ---------------------------------------------- */
		case ST_STEP_BEGIN: {
			if (L.d >= L.qlen) { L.state = ST_FELL_OFF; break; }
			const uint32_t d = L.d, cur = L.qlen - d - 1u;
			if (L.halfAndHalf && !bt_hh_check_top(L, S, d)) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
			if (L.ebase + (d - L.depth) >= S.entCap) { L.status |= BT_STF_OVERFLOW; bt_lane_finish(L, B); return; }
			const uint32_t c = bt_qry(L, cur), q = bt_qual(L, cur);
			L.c = c; L.q = q;
			const bool alt = (d >= L.fu) && (!L.considerQuals || (L.ham + bt_mm_penalty(L.maq, q) <= L.qualThresh));
			bool elig = false, over = false;
			if (alt) {
				if (L.considerQuals) {
					if (q < L.lowAltQual) { elig = true; over = true; }
					else if (q == L.lowAltQual) elig = true;
				} else elig = true;
			}
			L.fl_alt = alt; L.fl_elig = elig; L.fl_over = over;
			const uint32_t rtop = L.top, rbot = L.bot;
			if (c == 4u && d > 0) { L.top = 1; L.bot = 1; }
			if (rtop == 0 && rbot == 0) {
				/* depth 0: the fchr quartet (:531-543) */
				const uint32_t e = L.ebase + (d - L.depth);
				PT(e, 0) = ix.fchr[0];
				PB(e, 0) = PT(e, 1) = ix.fchr[1];
				PB(e, 1) = PT(e, 2) = ix.fchr[2];
				PB(e, 2) = PT(e, 3) = ix.fchr[3];
				PB(e, 3) = ix.fchr[4];
				if (c < 4u) { L.top = ix.fchr[c]; L.bot = ix.fchr[c + 1u]; }
				L.state = ST_STEP_POST;
			} else if (alt) {
				req.rowA = rtop; req.rowB = rbot; req.op = 3; L.lfk = LFK_EX2; L.cnt.lfex++;
				if (rtop / 448u == rbot / 448u) L.cnt.samePair++;
				L.state = ST_STEP_LFDONE; return;
			} else if (c < 4u) {
				if (L.top + 1u == L.bot) {
					req.rowA = L.top; req.op = 1; L.lfk = LFK_LF1; L.cnt.lf1++;
				} else {
					req.rowA = L.top; req.rowB = L.bot; req.op = 3; L.lfk = LFK_C2; L.cnt.lf2++;
					if (L.top / 448u == L.bot / 448u) L.cnt.samePair++;
				}
				L.state = ST_STEP_LFDONE; return;
			} else {
				L.state = ST_STEP_POST;
			}
			break;
		}

		case ST_STEP_LFDONE: {
			const uint32_t c = L.c;
			if (L.lfk == LFK_EX2) {
				const uint32_t e = L.ebase + (L.d - L.depth);
				PT(e, 0) = res.a[0]; PT(e, 1) = res.a[1]; PT(e, 2) = res.a[2]; PT(e, 3) = res.a[3];
				PB(e, 0) = res.b[0]; PB(e, 1) = res.b[1]; PB(e, 2) = res.b[2]; PB(e, 3) = res.b[3];
				if (c < 4u) { L.top = res.a[c]; L.bot = res.b[c]; }
			} else if (L.lfk == LFK_C2) {
				L.top = res.a[c]; L.bot = res.b[c];
			} else {
				/* mapLF1 (ebwt.h:2494-2512) */
				if (res.LA != c || L.top == ix.zOff) { L.top = L.bot = BT_OFF_MASK; }
				else { L.top = res.a[c]; L.bot = L.top + 1u; }
			}
			L.state = ST_STEP_POST;
			/* fallthrough */
		}
		// fallthrough
		case ST_STEP_POST: {
			const uint32_t d = L.d, cur = L.qlen - d - 1u, c = L.c, q = L.q;
			const uint32_t e = L.ebase + (d - L.depth);
			uint32_t el = (c < 4u) ? (1u << c) : 0u;
			if (L.fl_alt) {
				bool over = L.fl_over != 0;
				for (uint32_t i = 0; i < 4u; i++) {
					if (i == c) continue;
					uint32_t spread = PB(e, i) - PT(e, i);
					if (spread == 0) el |= (1u << i);
					else {
						if (L.fl_elig) {
							if (over) {
								L.lowAltQual = q; L.eligibleNum = 0; L.eligibleSz = 0; over = false;
								L.eli = d; L.eltop = PT(e, i); L.elbot = PB(e, i);
								L.elham = bt_mm_penalty(L.maq, q); L.elcint = i; L.elignore = 0;
							}
							L.eligibleSz += spread; L.eligibleNum++;
						}
						L.altNum++;
					}
				}
			}
			S.elims[e] = (uint8_t)el;
			bool btDespite = false, reportedPartial = false;
			if (cur == 0 && L.top < L.bot && L.sd < L.reportPartials && L.reportPartials > 0) {
				if (L.altNum > 0) btDespite = true;
				if (L.sd > 0) { bt_report_partial(L, S, L.sd); reportedPartial = true; }
			}
			bool invalidExact = false;
			if (cur == 0 && L.sd == 0 && L.bot > L.top && !L.reportExacts) { invalidExact = true; btDespite = true; }
			bool mustBacktrack = false, invalidHH = false;
			if (L.halfAndHalf) {
				if (d == L.d5 - 1u && L.top < L.bot) {
					invalidHH = (L.sd == 0);
					if (L.sd == 0 && L.altNum > 0) { btDespite = true; mustBacktrack = true; }
					else if (L.sd == 0) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
				} else if (d == L.d3 - 1u && L.top < L.bot) {
					uint32_t lo = 0, hi = 0;
					for (uint32_t i = 0; i < L.sd; i++) {
						uint32_t dd = L.qlen - (FRW(i, FR_MM) & 0xffffu) - 1u;
						if (dd < L.d5) hi++; else if (dd < L.d3) lo++;
					}
					invalidHH = (lo == 0 || hi == 0);
					if ((L.sd < 2 || invalidHH) && L.altNum > 0) { mustBacktrack = true; btDespite = true; }
					else if (L.sd < 2) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
				}
			}
			L.btDespite = btDespite;
			if (cur == 0 && L.bot > L.top && !invalidHH && !invalidExact && !reportedPartial) {
				BT_GOTO_RA(L.sd, L.top, L.bot, L.ham, RC_STEP);
				break;
			}
			if ((L.top == L.bot || btDespite) && L.altNum > 0) { L.state = ST_BT_LOOP; break; }
			if (mustBacktrack || invalidHH || invalidExact) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
			if (L.top == L.bot) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
			L.d = d + 1u;
			L.state = ST_STEP_BEGIN;
			break;
		}

		/* ---- choose a backtrack target and descend (:743-971) --------------------------- */
		case ST_BT_LOOP: {
			uint32_t i = L.d, j = 0, bttop = 0, btbot = 0, btham = L.ham, btcint = 0;
			if (L.eligibleNum > 1 || L.elignore) {
				bool found = false;
				for (;; i--) {
					const uint32_t icur = L.qlen - i - 1u;
					const uint32_t qi = bt_qual(L, icur);
					const uint32_t e = L.ebase + (i - L.depth);
					const uint32_t el = S.elims[e];
					if ((qi == L.lowAltQual || !L.considerQuals) && el != 15u) {
						uint32_t posSz = 0;
						for (j = 0; j < 4u; j++) if ((el & (1u << j)) == 0) posSz += PB(e, j) - PT(e, j);
						uint32_t r = (posSz > 0) ? (bt_rnd_u32(L) % posSz) : 0u;
						for (j = 0; j < 4u; j++) {
							if ((el & (1u << j)) == 0) {
								uint32_t spread = PB(e, j) - PT(e, j);
								if (r < spread) {
									bttop = PT(e, j); btbot = PB(e, j);
									btham += bt_mm_penalty(L.maq, qi);
									btcint = j; found = true;
									break;
								}
								r -= spread;
							}
						}
						break;
					}
					if (i == L.depth) break;
				}
				if (!found) { L.status |= BT_STF_OVERFLOW; bt_lane_finish(L, B); return; }   /* cannot happen */
			} else {
				i = L.eli; bttop = L.eltop; btbot = L.elbot; btham += L.elham; j = btcint = L.elcint;
			}
			const uint32_t icur = L.qlen - i - 1u;
			uint32_t nu = L.fu, n1 = L.f1, n2 = L.f2, n3 = L.f3;
			if (i < L.f1)      { nu = L.f1; n1 = L.f2; n2 = L.f3; }
			else if (i < L.f2) { n1 = L.f2; n2 = L.f3; }
			else if (i < L.f3) { n2 = L.f3; }
			FRW(L.sd, FR_MM) = icur | (btcint << 16);
			L.pi = i; L.pj = j; L.pbttop = bttop; L.pbtbot = btbot;
			if (i + 1u == L.qlen) {
				BT_GOTO_RA(L.sd + 1u, bttop, btbot, btham, RC_CHILD);
				break;
			}
			uint32_t newDepth = i + 1u, ntop = bttop, nbot = btbot;
			const bool rootNoFtab = (L.sd == 0) && L.nsFtab0;
			if (L.halfAndHalf && !rootNoFtab && L.r2 == L.r3 && i + 1u < ix.ftabChars && ix.ftabChars <= L.d5) {
				/* re-jump through the ftab with the substituted character (:908-952) */
				const uint32_t ftabChars = ix.ftabChars;
				uint32_t ftabOff = bt_qry(L, L.qlen - ftabChars);
				for (uint32_t jj = ftabChars - 1u; jj > 0; jj--) {
					ftabOff <<= 2;
					if (L.qlen - jj == icur) ftabOff |= btcint;
					else ftabOff |= bt_qry(L, L.qlen - jj);
				}
				ntop = bt_ftab_hi(ix, ftabOff); nbot = bt_ftab_lo(ix, ftabOff + 1u);
				L.cnt.ftab++;
				if (ntop == nbot) { L.ret = 0; L.state = ST_CHILD_RET; break; }
				newDepth = ftabChars;
			}
			/* push: save the parent, enter the child */
			if (L.sd + 1u >= S.frCap) { L.status |= BT_STF_OVERFLOW; bt_lane_finish(L, B); return; }
			{
				const uint32_t f = L.sd;
				FRW(f, FR_DEPTH) = L.depth; FRW(f, FR_D) = L.d; FRW(f, FR_HAM) = L.ham;
				FRW(f, FR_U) = L.fu; FRW(f, FR_R1) = L.f1; FRW(f, FR_R2) = L.f2; FRW(f, FR_R3) = L.f3;
				FRW(f, FR_ALTNUM) = L.altNum; FRW(f, FR_ELIGNUM) = L.eligibleNum; FRW(f, FR_ELIGSZ) = L.eligibleSz;
				FRW(f, FR_ELI) = L.eli; FRW(f, FR_ELTOP) = L.eltop; FRW(f, FR_ELBOT) = L.elbot;
				FRW(f, FR_ELHAM) = L.elham; FRW(f, FR_ELC) = L.elcint | (L.elignore << 8);
				FRW(f, FR_LOWQ) = L.lowAltQual;
				FRW(f, FR_PI) = L.pi; FRW(f, FR_PJ) = L.pj; FRW(f, FR_PTOP) = L.pbttop; FRW(f, FR_PBOT) = L.pbtbot;
				FRW(f, FR_EBASE) = L.ebase;
			}
			L.ebase = L.ebase + (L.d - L.depth + 1u);
			L.sd = L.sd + 1u; L.depth = newDepth; L.top = ntop; L.bot = nbot; L.ham = btham;
			L.fu = nu; L.f1 = n1; L.f2 = n2; L.f3 = n3;
			L.state = ST_FRAME_ENTER;
			break;
		}

		/* ---- a child frame (or a leaf report) came back (:972-1064) ---------------------- */
		case ST_CHILD_RET: {
			if (L.ret) { L.state = ST_FRAME_RETURN; break; }
			if (L.bailed || (L.halfAndHalf && L.maxBts > 0 && L.numBts >= L.maxBts)) {
				L.bailed = 1; L.ret = 0; L.state = ST_FRAME_RETURN; break;
			}
			{
				const uint32_t e = L.ebase + (L.pi - L.depth);
				S.elims[e] = (uint8_t)(S.elims[e] | (1u << L.pj));
			}
			L.eligibleSz -= (L.pbtbot - L.pbttop);
			L.eligibleNum--;
			L.elignore = 1;
			L.altNum--;
			if (L.altNum == 0) { L.ret = 0; L.state = ST_FRAME_RETURN; break; }
			if (L.eligibleNum == 0 && L.considerQuals) {
				/* re-scan the frame for the next-lowest-quality set of targets (:1004-1058) */
				L.lowAltQual = 0xff;
				for (uint32_t k = L.d; k >= L.depth && k <= L.qlen; k--) {
					const uint32_t kcur = L.qlen - k - 1u;
					const uint32_t kq = bt_qual(L, kcur);
					if (k < L.fu) break;
					const bool kAlt = (L.ham + bt_mm_penalty(L.maq, kq) <= L.qualThresh);
					bool kOver = false;
					if (kAlt) {
						if (kq < L.lowAltQual) kOver = true;
						if (kq <= L.lowAltQual) {
							const uint32_t e = L.ebase + (k - L.depth);
							const uint32_t el = S.elims[e];
							for (uint32_t l = 0; l < 4u; l++) {
								if ((el & (1u << l)) == 0) {
									uint32_t spread = PB(e, l) - PT(e, l);
									if (kOver) {
										L.lowAltQual = kq; kOver = false; L.eligibleNum = 0; L.eligibleSz = 0;
										L.eli = k; L.eltop = PT(e, l); L.elbot = PB(e, l);
										L.elham = bt_mm_penalty(L.maq, kq); L.elcint = l; L.elignore = 0;
									}
									L.eligibleNum++; L.eligibleSz += spread;
								}
							}
						}
					}
					if (k == 0) break;
				}
			}
			L.state = ST_BT_LOOP;
			break;
		}

		/* ---- return from a frame -------------------------------------------------------- */
		case ST_FRAME_RETURN: {
			if (L.sd == 0) { L.state = ST_SEARCH_END; break; }
			const uint32_t f = L.sd - 1u;
			L.sd = f;
			L.depth = FRW(f, FR_DEPTH); L.d = FRW(f, FR_D); L.ham = FRW(f, FR_HAM);
			L.fu = FRW(f, FR_U); L.f1 = FRW(f, FR_R1); L.f2 = FRW(f, FR_R2); L.f3 = FRW(f, FR_R3);
			L.altNum = FRW(f, FR_ALTNUM); L.eligibleNum = FRW(f, FR_ELIGNUM); L.eligibleSz = FRW(f, FR_ELIGSZ);
			L.eli = FRW(f, FR_ELI); L.eltop = FRW(f, FR_ELTOP); L.elbot = FRW(f, FR_ELBOT);
			L.elham = FRW(f, FR_ELHAM);
			{ uint32_t v = FRW(f, FR_ELC); L.elcint = v & 0xffu; L.elignore = (v >> 8) & 1u; }
			L.lowAltQual = FRW(f, FR_LOWQ);
			L.pi = FRW(f, FR_PI); L.pj = FRW(f, FR_PJ); L.pbttop = FRW(f, FR_PTOP); L.pbtbot = FRW(f, FR_PBOT);
			L.ebase = FRW(f, FR_EBASE);
			L.state = ST_CHILD_RET;
			break;
		}

		/* ---- ran off the 5' end of the query (:1086-1090) ------------------------------- */
		case ST_FELL_OFF: {
			if (L.sd >= L.reportPartials) BT_GOTO_RA(L.sd, L.top, L.bot, L.ham, RC_FELL);
			else { L.ret = 0; L.state = ST_FRAME_RETURN; }
			break;
		}

		/* ---- reportAlignment / reportFullAlignment (:1455-1565) -------------------------- */
		case ST_RA_BEGIN: {
			if (L.reportPartials) {
				if (L.ra_sd > 0) bt_report_partial(L, S, L.ra_sd);
				L.ret = 0; L.state = ST_RA_END; break;
			}
			uint32_t stratum = 0;
			for (uint32_t i = 0; i < L.ra_sd; i++)                      /* calcStratum (:1164-1177) */
				if ((FRW(i, FR_MM) & 0xffffu) >= (L.qlen - L.r3)) stratum++;
			stratum += L.nmuts;
			L.ra_nmm = L.ra_sd + L.nmuts;
			L.ra_stratum = stratum;
			L.ra_cost = (L.ra_cost | (stratum << 14)) & 0xffffu;
			if (L.ra_nmm == 0 && !L.reportExacts) { L.ret = 0; L.state = ST_RA_END; break; }
			{
				const uint32_t spread = L.ra_bot - L.ra_top;
				L.ra_r = L.ra_top + (bt_rnd_u32(L) % spread);
				L.ra_i = 0;
			}
			L.state = ST_ROW_BEGIN;
			break;
		}
		case ST_ROW_BEGIN: {
			const uint32_t spread = L.ra_bot - L.ra_top;
			if (L.ra_i >= spread) { L.ret = 0; L.state = ST_RA_END; break; }
			uint32_t ri = L.ra_r + L.ra_i;
			if (ri >= L.ra_bot) ri -= spread;
			L.crow = ri; L.cjumps = 0;
			L.state = ST_CHASE_CHECK;
			/* fallthrough */
		}
		// fallthrough
		case ST_CHASE_CHECK: {
			/* reportChaseOne's walk (ebwt.h:2727-2746) */
			if ((L.crow & ix.offMask) != L.crow && L.crow != ix.zOff) {
				req.rowA = L.crow; req.op = 1; L.cnt.chase++;
				L.state = ST_CHASE_LFDONE; return;
			}
			L.state = ST_RESOLVE;
			break;
		}
		case ST_CHASE_LFDONE: {
			L.crow = res.a[res.LA];                 /* mapLF(l), ebwt.h:2420 */
			L.cjumps++;
			L.state = ST_CHASE_CHECK;
			break;
		}
		case ST_RESOLVE: {
			uint32_t off;
			if (L.crow == ix.zOff) off = L.cjumps;
			else { off = ix.offs[L.crow >> ix.offRate] + L.cjumps; }
			L.cnt.offs++;
			uint32_t tidx = 0, toff = 0;
			if (bt_joined_to_text(ix, L.qlen, off, &tidx, &toff, &L.cnt.rstarts)) {
				if (bt_report_hit(L, P, ix, S, B, tidx, toff)) { L.ret = 1; L.state = ST_RA_END; break; }
			}
			L.ra_i++;
			L.state = ST_ROW_BEGIN;
			break;
		}
		case ST_RA_END: {
			switch (L.ra_cont) {
			case RC_STEP:
				if (L.ret) { L.state = ST_FRAME_RETURN; break; }
				L.top = L.bot;                                   /* keep looking (:730-735) */
				if (L.altNum > 0) L.state = ST_BT_LOOP;
				else { L.ret = 0; L.state = ST_FRAME_RETURN; }
				break;
			case RC_CHILD: L.state = ST_CHILD_RET; break;
			case RC_FELL:  L.state = ST_FRAME_RETURN; break;
			default:       L.state = ST_SEARCH_END; break;
			}
			break;
		}

		/* ---- backtrack() exit (:333-353, 303-324) + the seedling-extension loop ---------- */
		case ST_SEARCH_END: {
			L.numBts = 0;
			if (L.kind == BT_KIND_EXTEND) {
				/* search_seeded_phase3.c:9-59 / phase4.c:9-55: for each seedling, setMuts +
				 * backtrack(oldQuals); the RNG runs on across seedlings */
				if (L.palIdx != 0xffffffffu && L.ret) { bt_lane_finish(L, B); return; }
				L.palIdx = (L.palIdx == 0xffffffffu) ? 0u : L.palIdx + 1u;
				if (L.palIdx >= L.npals) { L.nmuts = 0; L.state = ST_PHASE_NEXT; break; }
				/* PartialAlignmentManager::toMutsString (ebwt_search_util.h:310-362) */
				const uint64_t pal = S.pals[L.palIdx];
				uint32_t pos[3] = { (uint32_t)(pal & 0xffffu), (uint32_t)((pal >> 16) & 0xffffu), (uint32_t)((pal >> 32) & 0xffffu) };
				uint32_t chr[3] = { (uint32_t)((pal >> 48) & 3u), (uint32_t)((pal >> 50) & 3u), (uint32_t)((pal >> 52) & 3u) };
				L.nmuts = 0;
				uint32_t oldQuals = 0;
				for (int k = 0; k < 3; k++) {
					if (k > 0 && pos[k] == 0xffffu) break;
					uint32_t tpos = L.plen - 1u - pos[k];
					oldQuals = (oldQuals + bt_mm_penalty(L.maq, bt_qual(L, tpos))) & 0xffu;
					L.mutpos[k] = tpos; L.mutnew[k] = chr[k];
					L.nmuts = (uint32_t)k + 1u;
				}
				L.iham = oldQuals;
				L.state = ST_SEARCH_BEGIN;
				break;
			}
			if (L.kind == BT_KIND_GEN) { L.state = ST_PHASE_NEXT; break; }
			if (L.ret) { bt_lane_finish(L, B); return; }
			L.state = ST_PHASE_NEXT;
			break;
		}
		default:
			return;
		}
	}
}

#undef FRW
#undef PT
#undef PB
#endif /* BT_CORE_H_ */
