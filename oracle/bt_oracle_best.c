/*
 * bt_oracle_best.c -- TEST INFRASTRUCTURE ONLY (see bt_oracle.h).
 *
 * CPU restatement of the reference's *stateful* (best-first) single-end search path, the one
 * `--best`, `--strata`, `-M` and `-v 3` select (ebwt_search.cpp:775-776, 851-853, 877-882):
 *
 *   UnpairedAlignerV2::setQuery/advance                 aligner.h:434-567
 *   RangeChaser / RowChaser                              range_chaser.h:52-209, row_chaser.h:69-123
 *   CostAwareRangeSourceDriver                           range_source.h:2033-2463
 *   EbwtSeededRangeSourceDriver                          ebwt_search_backtrack.h:2935-3141
 *   EbwtRangeSourceDriver (+SingleRangeSourceDriver)     ebwt_search_backtrack.h:2670-2839, range_source.h:1716-1882
 *   EbwtRangeSource::initBranch/advanceBranch            ebwt_search_backtrack.h:1921-2361
 *   PathManager / BranchQueue / CostCompare              range_source.h:1103-1574
 *   Branch / RangeState                                  range_source.h:314-1098
 *   driver trees per mode                                aligner_0mm.h:69-115, aligner_1mm.h:73-152,
 *                                                        aligner_23mm.h:73-236, aligner_seed_mm.h:82-516
 *   sinks                                                hit.h:969-985, 1070-1129, 1201-1209
 *
 * It follows the reference's object structure on purpose (one struct per class, one function
 * per method) so that each piece can be read against the lines it cites.  The product's HIP
 * implementation (bowtie_amd/csrc/bt_best.h) is an arena/index based automaton and shares no
 * code with this file.
 *
 * Things the reference leaves to its environment and that are pinned here:
 *   * std::priority_queue is libstdc++'s binary heap; Branch keys change while a Branch sits in
 *     the heap (curtail without cost change, extend), so the exact sift order matters.  heap_push /
 *     heap_pop below are libstdc++'s __push_heap / __adjust_heap (bits/stl_heap.h).
 *   * Branch ids come from AllocOnlyPool<Branch>::lastId() = (curPool_<<16)|cur_ with
 *     lim_ = 256 KB / sizeof(Branch) = 262144/136 = 1927 (pool.h:198,216-223,279-292,320-322).
 *   * ChunkPool exhaustion (--chunkmbs 64 MB per thread) is not modelled: a read that would
 *     exhaust it gets BT_ST_OVERFLOW here.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdlib.h>
#include "bt_oracle.h"

#define OFF_MASK 0xffffffffu
#define ADV_FOUND_RANGE  1
#define ADV_COST_CHANGES 2
#define ADV_STEP         3
#define BPOOL_LIM        1927u      /* chunkSz(256 KB) / sizeof(Branch)(136) */
#define BPOOL_LIM_WIDE   1638u      /* ... / sizeof(Branch)(160) in the BOWTIE_64BIT_INDEX build (bto_index.wide) */
#define BRANCH_BUDGET    400000     /* stand-in for ChunkPool exhaustion */

/* qual.h:15, qual.cpp:4-32 */
static uint8_t phred_of(uint8_t c) { return c >= 33 ? (uint8_t)(c - 33) : 0; }
static uint8_t mm_penalty(int maq, uint8_t q)
{
	if (!maq) return q;
	if (q < 5) return 0;
	if (q < 15) return 10;
	if (q < 25) return 20;
	return 30;
}

/* RandomSource::nextU32, random_source.h:45-54 */
static uint32_t rnd_next(uint32_t* last)
{
	uint32_t ret;
	*last = 1664525u * *last + 1013904223u;
	ret = *last >> 16;
	*last = 1664525u * *last + 1013904223u;
	ret ^= *last;
	return ret;
}

/* ------------------------------------------------------------------------------------------
 * Range (range.h:17-52), Edit (edit.h:82-86), RangeState (range_source.h:314-509),
 * Branch (range_source.h:517-1098)
 * ---------------------------------------------------------------------------------------- */
#define MAXE (BTO_MAXLEN + 16)

typedef struct {
	uint32_t top, bot;          /* top == OFF_MASK: invalid */
	uint16_t cost;
	uint32_t stratum, numMms;
	int fw, mate1;
	const bto_index* ebwt;
	uint32_t n;                 /* mms.size() */
	uint32_t mms[MAXE];
	uint8_t  refcs[MAXE];       /* ASCII */
} range_t;

typedef struct { uint16_t pos; uint8_t chr; } edit_t;

typedef struct {
	uint32_t tops[4], bots[4];
	uint8_t  mm[4];             /* eq.flags.mmA..mmT: 1 = eliminated */
	uint8_t  quallo;
	uint8_t  eliminated;
} rstate_t;

typedef struct branch {
	uint32_t id;
	uint16_t depth0, depth1, depth2, depth3, rdepth, len, cost, ham;
	rstate_t* ranges; uint16_t rangesSz;
	uint32_t top, bot;
	/* SideLocus ltop_/lbot_: only which rows they were initialised from matters */
	int ltopValid, lbotValid; uint32_t ltopRow, lbotRow;
	edit_t* edits; uint32_t nedits;
	uint16_t delayedCost;
	int curtailed, exhausted, delayedIncrease;
} branch_t;

typedef struct {
	int* budget;                /* per-read allocation budget (ChunkPool stand-in) */
	bt_op_counts* cnt;
} env_t;

/* Branch::prep / the locus part of Branch::init (range_source.h:559-566, 946-954) */
static void br_prep(branch_t* b)
{
	if (b->bot > b->top + 1) {
		b->ltopValid = b->lbotValid = 1; b->ltopRow = b->top; b->lbotRow = b->bot;
	} else if (b->bot > b->top) {
		b->ltopValid = 1; b->ltopRow = b->top; b->lbotValid = 0;
	}
}

/* Branch::eliminated, range_source.h:617-634 */
static int br_eliminated(const branch_t* b, int i)
{
	if (i <= (int)b->len && i < (int)b->rangesSz) return b->ranges[i].eliminated;
	return 1;
}

static void br_free(branch_t* b)
{
	free(b->ranges); free(b->edits); free(b);
}

/* Branch::init, range_source.h:527-604 */
static branch_t* br_new(env_t* env, uint32_t id, uint32_t qlen, uint16_t d0, uint16_t d1, uint16_t d2,
                        uint16_t d3, uint16_t rdepth, uint16_t len, uint16_t cost, uint16_t ham,
                        uint32_t itop, uint32_t ibot, const edit_t* edits, uint32_t nedits, int extraEdit)
{
	if (--(*env->budget) < 0) return NULL;
	if (env->cnt) env->cnt->frames++;
	branch_t* b = (branch_t*)calloc(1, sizeof(branch_t));
	b->id = id; b->depth0 = d0; b->depth1 = d1; b->depth2 = d2; b->depth3 = d3;
	b->rdepth = rdepth; b->len = len; b->cost = cost; b->ham = ham; b->top = itop; b->bot = ibot;
	br_prep(b);
	if (qlen > rdepth) {
		b->rangesSz = (uint16_t)(qlen - rdepth);
		b->ranges = (rstate_t*)calloc(b->rangesSz, sizeof(rstate_t));
	}
	b->edits = (edit_t*)malloc(sizeof(edit_t) * (nedits + (uint32_t)extraEdit + 1));
	if (nedits) memcpy(b->edits, edits, sizeof(edit_t) * nedits);
	b->nedits = nedits;
	for (uint32_t i = 0; i < len; i++) b->ranges[i].eliminated = 1;
	return b;
}

/* RangeState::pickEdit (mismatches only), range_source.h:321-485 */
static edit_t rs_pick_edit(rstate_t* r, int pos, uint32_t* rnd, uint32_t* top, uint32_t* bot, int* last)
{
	edit_t e; e.pos = (uint16_t)pos; e.chr = 0;
	int num = !r->mm[0] + !r->mm[1] + !r->mm[2] + !r->mm[3];
	if (num > 1) {
		*last = 0;
		uint32_t tot = 0;
		for (int c = 0; c < 4; c++) if (!r->mm[c]) tot += r->bots[c] - r->tops[c];
		uint32_t dart = rnd_next(rnd) % tot;
		for (int c = 0; c < 4; c++) {
			if (r->mm[c]) continue;
			if (c == 3 || dart < r->bots[c] - r->tops[c]) {
				*top = r->tops[c]; *bot = r->bots[c]; r->mm[c] = 1; e.chr = (uint8_t)"ACGT"[c];
				return e;
			}
			dart -= r->bots[c] - r->tops[c];
		}
	} else {
		*last = 1;
		int chr = !r->mm[0] ? 0 : !r->mm[1] ? 1 : !r->mm[2] ? 2 : 3;
		e.chr = (uint8_t)"ACGT"[chr];
		*top = r->tops[chr]; *bot = r->bots[chr];
		r->eliminated = 1;
	}
	return e;
}

/* Branch::installRanges, range_source.h:970-1023 (all four qualities are equal, :981-983) */
static int br_install_ranges(branch_t* b, int c, uint32_t qAllow, uint8_t q)
{
	rstate_t* r = &b->ranges[b->len];
	int ret = 0;
	r->eliminated = 1;
	r->mm[0] = r->mm[1] = r->mm[2] = r->mm[3] = 1;
	r->quallo = q;
	if (q > qAllow) return 0;
	for (int k = 0; k < 4; k++) {
		if (c != k && r->bots[k] > r->tops[k]) { r->eliminated = 0; r->mm[k] = 0; ret++; }
	}
	return ret;
}

/* Branch::curtail, range_source.h:877-939 */
static void br_curtail(branch_t* b, int seedLen, int qualOrder)
{
	if (b->ranges == NULL) { b->exhausted = 1; b->curtailed = 1; return; }
	uint16_t lowestCost = 0xffff;
	int i = (int)b->depth0 - (int)b->rdepth;
	if (i < 0) i = 0;
	for (; i <= (int)b->len; i++) {
		if (!br_eliminated(b, i)) {
			uint16_t stratum = ((int)b->rdepth + i < seedLen) ? (1 << 14) : 0;
			uint16_t cost = (uint16_t)((qualOrder ? b->ranges[i].quallo : 0) | stratum);
			if (cost < lowestCost) lowestCost = cost;
		}
	}
	if (lowestCost > 0 && lowestCost != 0xffff) b->cost = (uint16_t)(b->cost + lowestCost);
	else if (lowestCost == 0xffff) b->exhausted = 1;
	/* the trimming of ranges_ (:924-937) only returns memory to the pool */
	b->curtailed = 1;
}

/* ------------------------------------------------------------------------------------------
 * CostCompare + BranchQueue (range_source.h:1103-1283) on libstdc++'s heap, PathManager
 * (range_source.h:1307-1574)
 * ---------------------------------------------------------------------------------------- */
static int cost_compare(const branch_t* a, const branch_t* b)   /* true -> b before a */
{
	int aUn = a->curtailed || a->exhausted, bUn = b->curtailed || b->exhausted;
	if (a->cost == b->cost) {
		if (bUn && !aUn) return 0;
		if (aUn && !bUn) return 1;
		uint16_t ta = (uint16_t)(a->rdepth + a->len), tb = (uint16_t)(b->rdepth + b->len);
		if (ta != tb) return ta < tb;
		return b->id < a->id;
	}
	return b->cost < a->cost;
}

typedef struct {
	branch_t** q; int sz, cap;
	uint32_t bpCur, bpPool;          /* AllocOnlyPool<Branch>::cur_, curPool_ */
	uint32_t bpLim;                  /* its lim_ (0 = BPOOL_LIM) */
	uint32_t lastCur[64];            /* lastCurInPool_ */
	uint16_t minCost;
	int* btCnt;
	env_t* env;
} pathman_t;

static void heap_push(pathman_t* pm, branch_t* v)      /* std::push_heap */
{
	if (pm->sz == pm->cap) { pm->cap = pm->cap ? pm->cap * 2 : 16; pm->q = (branch_t**)realloc(pm->q, sizeof(branch_t*) * (size_t)pm->cap); }
	int hole = pm->sz++;
	int parent = (hole - 1) / 2;
	while (hole > 0 && cost_compare(pm->q[parent], v)) {
		pm->q[hole] = pm->q[parent];
		hole = parent;
		parent = (hole - 1) / 2;
	}
	pm->q[hole] = v;
}

static branch_t* heap_pop(pathman_t* pm)               /* std::pop_heap + pop_back */
{
	branch_t* top = pm->q[0];
	int n = pm->sz;
	if (n > 1) {
		branch_t* value = pm->q[n - 1];
		pm->q[n - 1] = pm->q[0];
		int len = n - 1, hole = 0, second = 0;
		while (second < (len - 1) / 2) {
			second = 2 * (second + 1);
			if (cost_compare(pm->q[second], pm->q[second - 1])) second--;
			pm->q[hole] = pm->q[second];
			hole = second;
		}
		if ((len & 1) == 0 && second == (len - 2) / 2) {
			second = 2 * (second + 1);
			pm->q[hole] = pm->q[second - 1];
			hole = second - 1;
		}
		int parent = (hole - 1) / 2;
		while (hole > 0 && cost_compare(pm->q[parent], value)) {
			pm->q[hole] = pm->q[parent];
			hole = parent;
			parent = (hole - 1) / 2;
		}
		pm->q[hole] = value;
	}
	pm->sz--;
	return top;
}

/* AllocOnlyPool<Branch>::alloc + lastId (pool.h:216-223, 320-322, 335-352) */
static uint32_t pm_alloc_id(pathman_t* pm)
{
	if (pm->bpCur + 1 >= (pm->bpLim ? pm->bpLim : BPOOL_LIM)) {
		if (pm->bpPool < 63) pm->lastCur[pm->bpPool] = pm->bpCur;
		pm->bpPool++; pm->bpCur = 0;
	}
	pm->bpCur++;
	return (pm->bpPool << 16) | pm->bpCur;
}
/* AllocOnlyPool<Branch>::free(T*) (pool.h:279-292): only the topmost element is given back */
static void pm_free_branch(pathman_t* pm, branch_t* b)
{
	if (pm->bpCur > 0 && b->id == ((pm->bpPool << 16) | pm->bpCur)) {
		pm->bpCur--;
		if (pm->bpCur == 0 && pm->bpPool > 0) { pm->bpPool--; pm->bpCur = pm->lastCur[pm->bpPool < 64 ? pm->bpPool : 63]; }
	}
	br_free(b);
}

static void pm_reset(pathman_t* pm)
{
	for (int i = 0; i < pm->sz; i++) br_free(pm->q[i]);
	pm->sz = 0; pm->bpCur = 0; pm->bpPool = 0; pm->minCost = 0;
}
static branch_t* pm_front(pathman_t* pm) { return pm->q[0]; }
/* PathManager::pop (range_source.h:1337-1356): minCost is read from branchQ_.front() even when
 * the queue just became empty -- vector::front() of an emptied vector is the element that was
 * just removed */
static branch_t* pm_pop(pathman_t* pm)
{
	branch_t* b = heap_pop(pm);
	pm->minCost = (pm->sz > 0) ? pm->q[0]->cost : b->cost;
	return b;
}
static void pm_push(pathman_t* pm, branch_t* b) { heap_push(pm, b); pm->minCost = pm->q[0]->cost; }

/* PathManager::curtail, range_source.h:1435-1454 */
static void pm_curtail(pathman_t* pm, branch_t* br, int seedLen, int qualOrder)
{
	uint16_t origCost = br->cost;
	br_curtail(br, seedLen, qualOrder);
	if (br->exhausted) { pm_pop(pm); pm_free_branch(pm, br); }
	else if (br->cost != origCost) { branch_t* p = pm_pop(pm); pm_push(pm, p); }
}

/* Branch::splitBranch, range_source.h:644-773 */
static branch_t* br_split(pathman_t* pm, branch_t* b, uint32_t* rnd, uint32_t qlen, int seedLen, int qualOrder)
{
	uint32_t id = pm_alloc_id(pm);
	int tied[3], numTied = 0, numNotElim = 0;
	uint16_t bestCost = 0xffff, nextCost = 0xffff;
	int i = (int)b->depth0 - (int)b->rdepth;
	if (i < 0) i = 0;
	for (; i <= (int)b->len; i++) {
		if (br_eliminated(b, i)) continue;
		numNotElim++;
		uint16_t stratum = ((int)b->rdepth + i < seedLen) ? (1 << 14) : 0;
		uint16_t cost = (uint16_t)(stratum | (qualOrder ? b->ranges[i].quallo : 0));
		if (cost < bestCost) { nextCost = bestCost; bestCost = cost; numTied = 1; tied[0] = i; }
		else if (cost == bestCost) {
			if (numTied < 3) tied[numTied++] = i;
			else { tied[0] = tied[1]; tied[1] = tied[2]; tied[2] = i; }
		} else if (cost < nextCost) nextCost = cost;
	}
	int r = 0;
	if (numTied > 1) r = (int)(rnd_next(rnd) % (uint32_t)numTied);
	int pos = tied[r];
	int last = 0;
	uint32_t top = 0, bot = 0;
	edit_t e = rs_pick_edit(&b->ranges[pos], pos + b->rdepth, rnd, &top, &bot, &last);
	uint16_t newRdepth = (uint16_t)(b->rdepth + pos + 1);
	uint32_t hamadd = bestCost & ~0xc000u;
	uint16_t depth = (uint16_t)(pos + b->rdepth);
	uint16_t nd0 = b->depth0, nd1 = b->depth1, nd2 = b->depth2, nd3 = b->depth3;
	if (depth < b->depth1) nd0 = b->depth1;
	if (depth < b->depth2) nd1 = b->depth2;
	if (depth < b->depth3) nd2 = b->depth3;
	branch_t* nb = br_new(pm->env, id, qlen, nd0, nd1, nd2, nd3, newRdepth, 0, b->cost,
	                      (uint16_t)(b->ham + hamadd), top, bot, b->edits, b->nedits, 1);
	if (!nb) return NULL;
	nb->edits[nb->nedits++] = e;
	if (numNotElim == 1 && last) b->exhausted = 1;
	else if (numTied == 1 && last) {
		if (bestCost != nextCost) {
			b->delayedCost = (uint16_t)(b->cost - bestCost + nextCost);
			b->delayedIncrease = 1;
		}
	}
	return nb;
}

/* PathManager::splitAndPrep, range_source.h:1460-1518 */
static int pm_split_and_prep(pathman_t* pm, uint32_t* rnd, uint32_t qlen, int seedLen, int qualOrder)
{
	if (pm->sz == 0) return 1;
	if (pm->btCnt && *pm->btCnt == 0) return 0;
	branch_t* f = pm_front(pm);
	while (f->delayedIncrease) {
		pm_pop(pm);
		f->cost = f->delayedCost; f->delayedIncrease = 0; f->delayedCost = 0;
		pm_push(pm, f);
		f = pm_front(pm);
	}
	if (f->curtailed) {
		if (pm->btCnt) { if (--(*pm->btCnt) == 0) return 0; }
		branch_t* nb = br_split(pm, f, rnd, qlen, seedLen, qualOrder);
		if (!nb) return 0;
		if (f->exhausted) { pm_pop(pm); pm_free_branch(pm, f); }
		pm_push(pm, nb);
	}
	if (pm->sz) br_prep(pm_front(pm));
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * The read in its four forms (Read::patFw/patRc/patFwRev/patRcRev, qual/qualRev; read.h)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t len, seed;
	uint8_t pat[2][2][BTO_MAXLEN];   /* [fw][ebwtFw] */
	uint8_t qual[2][BTO_MAXLEN];     /* [0] = qual, [1] = qualRev */
} read_t;

static void read_init(read_t* rd, const uint8_t* seq, const uint8_t* qual, uint32_t len, uint32_t seed)
{
	rd->len = len; rd->seed = seed;
	for (uint32_t i = 0; i < len; i++) {
		uint8_t c = seq[i], rc = seq[len - 1 - i];
		if (rc < 4) rc ^= 3;
		rd->pat[1][1][i] = c;                 /* patFw    */
		rd->pat[1][0][len - 1 - i] = c;       /* patFwRev */
		rd->pat[0][1][i] = rc;                /* patRc    */
		rd->pat[0][0][len - 1 - i] = rc;      /* patRcRev */
		rd->qual[0][i] = qual[i];
		rd->qual[1][len - 1 - i] = qual[i];
	}
}

/* ------------------------------------------------------------------------------------------
 * EbwtRangeSource, ebwt_search_backtrack.h:1788-2599
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	const bto_index* ebwt;
	int fw;
	uint32_t qualLim;
	int reportExacts, halfAndHalf, partial, maqPenalty, qualOrder;
	uint32_t offRev0, offRev1, offRev2, offRev3, depth5, depth3;
	const uint8_t *qry, *qual;
	uint8_t qryBuf[BTO_MAXLEN];
	uint32_t qlen;
	uint32_t rnd;
	range_t curRange, seedRange;
	int skippingThisRead, done, foundRange;
	env_t* env;
} rsrc_t;

/* EbwtRangeSource::setQuery, :1831-1870 */
static void rs_set_query(rsrc_t* s, const read_t* rd, const range_t* seedRange)
{
	int ebwtFw = s->ebwt->fw != 0;
	s->qry = rd->pat[s->fw][ebwtFw];
	s->qual = ebwtFw ? (s->fw ? rd->qual[0] : rd->qual[1]) : (s->fw ? rd->qual[1] : rd->qual[0]);
	if (seedRange) s->seedRange = *seedRange; else s->seedRange.top = OFF_MASK;
	s->qlen = rd->len;
	s->skippingThisRead = 0;
	if (s->seedRange.top != OFF_MASK) {
		memcpy(s->qryBuf, s->qry, rd->len);
		for (uint32_t i = 0; i < s->seedRange.n; i++) {
			uint8_t rc = s->seedRange.refcs[i];
			s->qryBuf[s->qlen - s->seedRange.mms[i] - 1] = (uint8_t)(rc == 'A' ? 0 : rc == 'C' ? 1 : rc == 'G' ? 2 : 3);
		}
		s->qry = s->qryBuf;
	}
	s->done = 0; s->foundRange = 0;
	s->rnd = rd->seed;
}

/* addPartialEdits, :2377-2387 */
static void rs_add_partial_edits(rsrc_t* s)
{
	if (s->seedRange.top == OFF_MASK) return;
	for (uint32_t i = 0; i < s->seedRange.n; i++) {
		s->curRange.mms[s->curRange.n] = s->qlen - s->seedRange.mms[i] - 1;
		s->curRange.refcs[s->curRange.n++] = s->seedRange.refcs[i];
	}
	s->curRange.numMms += s->seedRange.n;
}

/* tallyNs, :2490-2523 */
static int rs_tally_ns(const rsrc_t* s, int* nsInFtab)
{
	int nsInSeed = 0;
	for (uint32_t i = 0; i < s->offRev3; i++) {
		if (s->qry[s->qlen - i - 1] == 4) {
			nsInSeed++;
			if (nsInSeed == 1) { if (i < s->offRev0) return 0; }
			else if (nsInSeed == 2) { if (i < s->offRev1) return 0; }
			else if (nsInSeed == 3) { if (i < s->offRev2) return 0; }
			else return 0;
		}
	}
	for (uint32_t i = 0; i < s->ebwt->ftabChars && i < s->qlen; i++)
		if (s->qry[s->qlen - i - 1] == 4) (*nsInFtab)++;
	return 1;
}

/* initBranch, :1920-2051 */
static void rs_init_branch(rsrc_t* s, pathman_t* pm)
{
	const bto_index* ix = s->ebwt;
	uint32_t ftabChars = ix->ftabChars;
	s->foundRange = 0;
	int nsInFtab = 0;
	if (s->skippingThisRead) { s->done = 1; return; }
	if (s->qlen < 4) {
		uint32_t maxmms = 0;
		if (s->offRev0 != s->offRev1) maxmms = 1;
		if (s->offRev1 != s->offRev2) maxmms = 2;
		if (s->offRev2 != s->offRev3) maxmms = 3;
		if (s->qlen <= maxmms) { s->done = 1; s->skippingThisRead = 1; return; }
	}
	if (!rs_tally_ns(s, &nsInFtab)) return;
	int valid = s->seedRange.top != OFF_MASK;
	uint16_t icost = valid ? s->seedRange.cost : 0;
	uint16_t iham = (valid && s->qualOrder) ? (uint16_t)(s->seedRange.cost & ~0xc000) : 0;
	uint32_t m = s->offRev0 < s->qlen ? s->offRev0 : s->qlen;
	int ftabSkipsToEnd = (s->qlen == ftabChars);
	int skipInvalidExact = (!s->reportExacts && ftabSkipsToEnd);
	if (nsInFtab == 0 && m >= ftabChars && !skipInvalidExact) {
		/* calcFtabOff, :2530-2544 */
		uint32_t ftabOff = s->qry[s->qlen - ftabChars];
		for (int i = (int)ftabChars - 1; i > 0; i--) { ftabOff <<= 2; ftabOff |= s->qry[s->qlen - (uint32_t)i]; }
		uint32_t top = bto_ftab_hi(ix, ftabOff), bot = bto_ftab_lo(ix, ftabOff + 1);
		if (s->env->cnt) s->env->cnt->ftab++;
		if (s->qlen == ftabChars && bot > top) {
			range_t* r = &s->curRange;
			r->top = top; r->bot = bot; r->stratum = icost >> 14; r->cost = icost; r->numMms = 0;
			r->ebwt = ix; r->fw = s->fw; r->n = 0;
			rs_add_partial_edits(s);
			s->foundRange = 1;
			return;
		} else if (bot > top) {
			branch_t* b = br_new(s->env, pm_alloc_id(pm), s->qlen, (uint16_t)s->offRev0, (uint16_t)s->offRev1,
			                     (uint16_t)s->offRev2, (uint16_t)s->offRev3, 0, (uint16_t)ftabChars, icost, iham,
			                     top, bot, NULL, 0, 0);
			if (!b) return;
			pm_push(pm, b);
		}
	} else {
		branch_t* b = br_new(s->env, pm_alloc_id(pm), s->qlen, (uint16_t)s->offRev0, (uint16_t)s->offRev1,
		                     (uint16_t)s->offRev2, (uint16_t)s->offRev3, 0, 0, icost, iham, 0, 0, NULL, 0, 0);
		if (!b) return;
		pm_push(pm, b);
	}
}

/* hhCheckTop, :2444-2475; hhCheck, :2397-2436 */
static int rs_hh_check_top(const rsrc_t* s, const branch_t* b, uint32_t d)
{
	if (d == s->depth5) { if (b->nedits == 0) return 0; }
	else if (d == s->depth3) { if (b->nedits < (uint32_t)s->halfAndHalf) return 0; }
	return 1;
}
static int rs_hh_check(const rsrc_t* s, const branch_t* b, uint32_t depth, int empty)
{
	if (depth == s->depth5 - 1 && !empty) return b->nedits > 0;
	else if (depth == s->depth3 - 1 && !empty) {
		uint32_t lo = 0, hi = 0;
		for (uint32_t i = 0; i < b->nedits; i++) {
			uint32_t d = b->edits[i].pos;
			if (d < s->depth5) hi++; else if (d < s->depth3) lo++;
		}
		int invalid = (lo == 0 || hi == 0);
		return b->nedits >= (uint32_t)s->halfAndHalf && !invalid;
	}
	return 1;
}

/* mapLF1(row&, l) ebwt.h:2530; mapLF1(row, l, c) ebwt.h:2494; mapLF(l, c) ebwt.h:2458 */
static uint32_t lf_c(const bto_index* ix, uint32_t row, int c)
{
	uint32_t lf[4]; bto_rank4(ix, row, lf); return lf[c];
}

/* advanceBranch, :2059-2361 */
static void rs_advance_branch(rsrc_t* s, int until, pathman_t* pm)
{
	const bto_index* ix = s->ebwt;
	bt_op_counts* cnt = s->env->cnt;
	s->foundRange = 0;
	do {
		branch_t* br = pm_front(pm);
		uint32_t depth = (uint32_t)br->rdepth + br->len;
		int empty, hit, invalidExact;
		uint16_t cost = br->cost;
		uint32_t cur = 0, nedits;
		if (s->halfAndHalf && !rs_hh_check_top(s, br, depth)) {
			pm_curtail(pm, br, (int)s->depth3, s->qualOrder);
			goto bail;
		}
		cur = s->qlen - depth - 1;
		if (depth < s->qlen) {
			int c = s->qry[cur];
			uint8_t q = mm_penalty(s->maqPenalty, phred_of(s->qual[cur]));
			int curIsAlternative = (depth >= br->depth0) && ((uint32_t)br->ham + q <= s->qualLim);
			uint32_t otop = br->top;
			if (c == 4 && depth > 0) br->top = br->bot = 1;
			rstate_t* rs = &br->ranges[br->len];
			if (br->top == 0 && br->bot == 0) {
				rs->tops[0] = ix->fchr[0];
				rs->bots[0] = rs->tops[1] = ix->fchr[1];
				rs->bots[1] = rs->tops[2] = ix->fchr[2];
				rs->bots[2] = rs->tops[3] = ix->fchr[3];
				rs->bots[3] = ix->fchr[4];
				br_install_ranges(br, c, s->qualLim - br->ham, q);
				if (c < 4) { br->top = rs->tops[c]; br->bot = rs->bots[c]; }
			} else if (curIsAlternative && (br->bot > br->top || c == 4)) {
				for (int k = 0; k < 4; k++) rs->tops[k] = rs->bots[k] = 0;
				if (br->lbotValid) {
					bto_rank4(ix, br->ltopRow, rs->tops);
					bto_rank4(ix, br->lbotRow, rs->bots);
					if (cnt) { cnt->lfex++; if (br->ltopRow / 448 == br->lbotRow / 448) cnt->same_pair++; }
				} else {
					/* mapLF1(otop, ltop_) */
					if (cnt) cnt->lf1++;
					int cc = -1;
					if (otop != ix->zOff) { cc = bto_rowL(ix, br->ltopRow); otop = lf_c(ix, br->ltopRow, cc); }
					br->top = otop;
					if (cc >= 0) { rs->tops[cc] = br->top; rs->bots[cc] = br->top + 1; }
				}
				br_install_ranges(br, c, s->qualLim - br->ham, q);
				if (c < 4) { br->top = rs->tops[c]; br->bot = rs->bots[c]; }
				else br->top = br->bot = 1;
			} else if (br->bot > br->top) {
				rs->eliminated = 1;
				if (c < 4) {
					if (br->top + 1 == br->bot) {
						if (cnt) cnt->lf1++;
						/* mapLF1(top_, ltop_, c) */
						if (bto_rowL(ix, br->ltopRow) != c || br->top == ix->zOff) br->bot = br->top = OFF_MASK;
						else br->bot = br->top = lf_c(ix, br->ltopRow, c);
						if (br->bot != OFF_MASK) br->bot++;
					} else {
						if (cnt) { cnt->lf2++; if (br->ltopRow / 448 == br->lbotRow / 448) cnt->same_pair++; }
						br->top = lf_c(ix, br->ltopRow, c);
						br->bot = lf_c(ix, br->lbotRow, c);
					}
				}
			} else {
				rs->eliminated = 1;
			}
		} else {
			cur = 0;
		}
		empty = (br->top == br->bot);
		hit = (cur == 0 && !empty);
		nedits = br->nedits;
		invalidExact = (hit && nedits == 0 && !s->reportExacts);
		if (s->halfAndHalf && !rs_hh_check(s, br, depth, empty)) {
			pm_curtail(pm, br, (int)s->depth3, s->qualOrder);
			goto bail;
		}
		if (hit && !invalidExact) {
			range_t* r = &s->curRange;
			r->top = br->top; r->bot = br->bot; r->cost = br->cost; r->stratum = br->cost >> 14;
			r->numMms = nedits; r->fw = s->fw; r->n = 0;
			for (uint32_t i = 0; i < nedits; i++) {
				r->mms[r->n] = s->qlen - br->edits[i].pos - 1;
				r->refcs[r->n++] = br->edits[i].chr;
			}
			rs_add_partial_edits(s);
			r->ebwt = ix;
			s->foundRange = 1;
			pm_curtail(pm, br, (int)s->depth3, s->qualOrder);
		} else if (empty || cur == 0) {
			pm_curtail(pm, br, (int)s->depth3, s->qualOrder);
		} else {
			br->len++;                                   /* Branch::extend */
		}
	bail:
		if (!pm_split_and_prep(pm, &s->rnd, s->qlen, (int)s->depth3, s->qualOrder)) pm_reset(pm);
		if (pm->sz == 0) break;
		if (until == ADV_COST_CHANGES && pm_front(pm)->cost != cost) break;
		else if (until == ADV_STEP) break;
	} while (!s->foundRange);
}

/* ------------------------------------------------------------------------------------------
 * RangeSourceDrivers
 * ---------------------------------------------------------------------------------------- */
enum { PIN_TO_BEGINNING = 1, PIN_TO_LEN, PIN_TO_HI_HALF_EDGE, PIN_TO_SEED_EDGE };
enum { DR_SINGLE, DR_COST, DR_SEEDED };

typedef struct {                       /* EbwtRangeSourceDriverFactory + EbwtRangeSourceFactory */
	const bto_index* ebwt; int fw; uint32_t qualLim; int reportExacts, halfAndHalf, partial;
	int seed; uint32_t seedLen; int nudgeLeft; int rev[4]; int* btCnt;
	int mate1;
} single_spec_t;

typedef struct driver {
	int type;
	int foundRange, done;
	uint16_t minCost, minCostAdjustment;
	int fw, mate1;
	env_t* env; int maq, qualOrder;
	/* DR_SINGLE: EbwtRangeSourceDriver */
	rsrc_t* rs; pathman_t pm; uint32_t len;
	int seed; uint32_t seedLen; int nudgeLeft; int rev[4];
	/* DR_COST: CostAwareRangeSourceDriver */
	struct driver **rss, **active; int nrss, nactive, capRss;
	int paired, strandFix; uint32_t rnd;
	range_t *lastRange, *delayedRange;
	const read_t* patsrc;
	/* DR_SEEDED: EbwtSeededRangeSourceDriver */
	single_spec_t rsFact; struct driver *rsFull, *rsSeed; range_t* seedRange;
} driver_t;

static void     drv_set_query(driver_t* d, const read_t* rd, range_t* r);
static void     drv_advance(driver_t* d, int until);
static range_t* drv_range(driver_t* d);

static driver_t* single_new(env_t* env, int maq, int qualOrder, const single_spec_t* sp)
{
	driver_t* d = (driver_t*)calloc(1, sizeof(driver_t));
	d->type = DR_SINGLE; d->done = 1; d->env = env; d->maq = maq; d->qualOrder = qualOrder;
	d->fw = sp->fw; d->mate1 = sp->mate1;
	d->rs = (rsrc_t*)calloc(1, sizeof(rsrc_t));
	d->pm.bpLim = sp->ebwt->wide ? BPOOL_LIM_WIDE : BPOOL_LIM;
	d->rs->ebwt = sp->ebwt; d->rs->fw = sp->fw; d->rs->qualLim = sp->qualLim; d->rs->reportExacts = sp->reportExacts;
	d->rs->halfAndHalf = sp->halfAndHalf; d->rs->partial = sp->partial; d->rs->maqPenalty = maq; d->rs->qualOrder = qualOrder;
	d->rs->env = env; d->rs->curRange.top = OFF_MASK; d->rs->seedRange.top = OFF_MASK;
	d->seed = sp->seed; d->seedLen = sp->seedLen; d->nudgeLeft = sp->nudgeLeft;
	memcpy(d->rev, sp->rev, sizeof(d->rev));
	d->pm.btCnt = sp->btCnt; d->pm.env = env;
	return d;
}

static void drv_free(driver_t* d)
{
	if (!d) return;
	if (d->type == DR_SINGLE) { pm_reset(&d->pm); free(d->pm.q); free(d->rs); }
	else if (d->type == DR_COST) { for (int i = 0; i < d->nrss; i++) drv_free(d->rss[i]); free(d->rss); free(d->active); }
	else { drv_free(d->rsFull); drv_free(d->rsSeed); }
	free(d);
}

static uint32_t cext_to_depth(int cext, uint32_t sRight, uint32_t s, uint32_t len)
{
	if (cext == PIN_TO_SEED_EDGE) return s;
	if (cext == PIN_TO_HI_HALF_EDGE) return sRight;
	if (cext == PIN_TO_BEGINNING) return 0;
	return len;
}

/* EbwtRangeSourceDriver::initRangeSource, ebwt_search_backtrack.h:2721-2806 */
static void single_init_range_source(driver_t* d, const uint8_t* qual)
{
	rsrc_t* rs = d->rs;
	uint32_t len = d->len;
	uint32_t s = d->seedLen > 0 ? (d->seedLen < len ? d->seedLen : len) : len;
	uint32_t sRight = s >> 1;
	if ((s & 1) != 0 && !d->nudgeLeft) sRight++;
	uint32_t r0 = cext_to_depth(d->rev[0], sRight, s, len), r1 = cext_to_depth(d->rev[1], sRight, s, len);
	uint32_t r2 = cext_to_depth(d->rev[2], sRight, s, len), r3 = cext_to_depth(d->rev[3], sRight, s, len);
	uint32_t qlen = len;
	if (d->seed && len > s) { rs->qlen = rs->qlen < s ? rs->qlen : s; qlen = s; }
	uint16_t minCost = 0;
	if (rs->reportExacts) {
	} else if (!rs->halfAndHalf && r0 < s) {
		minCost = 1 << 14;
		if (d->qualOrder) {
			uint8_t low = 0xff;
			for (uint32_t k = r0; k < s; k++) { uint8_t c = qual[qlen - k - 1]; if (c < low) low = c; }
			minCost = (uint16_t)(minCost + mm_penalty(d->maq, phred_of(low)));
		}
	} else if (rs->halfAndHalf && sRight > 0 && sRight < (s - 1)) {
		minCost = (uint16_t)((d->seed ? 3 : 2) << 14);
		if (d->qualOrder) {
			uint8_t low1 = 0xff;
			for (uint32_t k = 0; k < sRight; k++) { uint8_t c = qual[qlen - k - 1]; if (c < low1) low1 = c; }
			minCost = (uint16_t)(minCost + mm_penalty(d->maq, phred_of(low1)));
			uint8_t l21 = 0xff, l22 = 0xff;
			for (uint32_t k = sRight; k < s; k++) {
				uint8_t c = qual[qlen - k - 1];
				if (c < l21) { if (l21 != 0xff) l22 = l21; l21 = c; }
				else if (c < l22) l22 = c;
			}
			minCost = (uint16_t)(minCost + mm_penalty(d->maq, phred_of(l21)));
			if (rs->halfAndHalf > 2 && l22 != 0xff) minCost = (uint16_t)(minCost + mm_penalty(d->maq, phred_of(l22)));
		}
	}
	d->minCostAdjustment = minCost;
	rs->depth5 = sRight; rs->depth3 = s; rs->offRev0 = r0; rs->offRev1 = r1; rs->offRev2 = r2; rs->offRev3 = r3;
}

/* SingleRangeSourceDriver::setQueryImpl, range_source.h:1750-1771 */
static void single_set_query(driver_t* d, const read_t* patsrc, range_t* r)
{
	const read_t* rd = &patsrc[d->mate1 ? 0 : 1];      /* bufa() / bufb() */
	d->done = 0;
	pm_reset(&d->pm);
	d->len = rd->len;
	rs_set_query(d->rs, rd, r);
	int ebwtFw = d->rs->ebwt->fw != 0;
	single_init_range_source(d, (d->fw == ebwtFw) ? rd->qual[0] : rd->qual[1]);
	if (!d->rs->done) rs_init_branch(d->rs, &d->pm);
	uint16_t icost = r ? r->cost : 0;
	d->minCost = icost > d->minCostAdjustment ? icost : d->minCostAdjustment;
	d->done = d->rs->done;
	d->foundRange = d->rs->foundRange;
}

/* SingleRangeSourceDriver::advanceImpl, range_source.h:1777-1838 */
static void single_advance(driver_t* d, int until)
{
	if (d->done || d->pm.sz == 0) { d->done = 1; return; }
	rs_advance_branch(d->rs, until, &d->pm);
	d->done = (d->pm.sz == 0);
	if (d->pm.minCost != 0) d->minCost = d->pm.minCost > d->minCostAdjustment ? d->pm.minCost : d->minCostAdjustment;
	d->foundRange = d->rs->foundRange;
}

/* ---- CostAwareRangeSourceDriver, range_source.h:2033-2463 ---- */
static driver_t* cost_new(env_t* env, int strandFix)
{
	driver_t* d = (driver_t*)calloc(1, sizeof(driver_t));
	d->type = DR_COST; d->env = env; d->strandFix = strandFix; d->mate1 = 1; d->fw = 1;
	return d;
}
static void cost_add_rss(driver_t* d, driver_t* p)
{
	if (d->nrss == d->capRss) {
		d->capRss = d->capRss ? d->capRss * 2 : 8;
		d->rss = (driver_t**)realloc(d->rss, sizeof(driver_t*) * (size_t)d->capRss);
		d->active = (driver_t**)realloc(d->active, sizeof(driver_t*) * (size_t)d->capRss);
	}
	d->rss[d->nrss++] = p;
}
/* calcPaired (range_source.h:2251-2261), mateEliminated (:2266-2280) */
static void cost_calc_paired(driver_t* d)
{
	int saw1 = 0, saw2 = 0;
	for (int i = 0; i < d->nrss; i++) { if (d->rss[i]->mate1) saw1 = 1; else saw2 = 1; }
	d->paired = saw1 && saw2;
}
static int cost_mate_eliminated(driver_t* d)
{
	if (!d->paired) return 0;
	int m1 = 0, m2 = 0;
	for (int i = 0; i < d->nactive; i++) if (!d->active[i]->done) { if (d->active[i]->mate1) m1 = 1; else m2 = 1; }
	return !m1 || !m2;
}

/* sortActives, :2370-2415 */
static void cost_sort_actives(driver_t* d)
{
	driver_t** vec = d->active;
	int sz = d->nactive;
	for (int i = 0; i < sz;) {
		if (vec[i]->done && !vec[i]->foundRange) {
			memmove(vec + i, vec + i + 1, sizeof(driver_t*) * (size_t)(d->nactive - i - 1));
			d->nactive--;
			if (sz == 0) break; else sz--;
			continue;
		}
		uint16_t minCost = vec[i]->minCost;
		int minOff = i;
		for (int j = i + 1; j < sz; j++) {
			if (vec[j]->done && !vec[j]->foundRange) continue;
			if (vec[j]->minCost < minCost) { minCost = vec[j]->minCost; minOff = j; }
			else if (vec[j]->minCost == minCost) { if (rnd_next(&d->rnd) & 0x1000) minOff = j; }
		}
		if (i != minOff) { driver_t* t = vec[i]; vec[i] = vec[minOff]; vec[minOff] = t; }
		i++;
	}
	if (d->delayedRange == NULL && sz > 0) d->minCost = vec[0]->minCost;
}

/* setQueryImpl, :2076-2093 */
static void cost_set_query(driver_t* d, const read_t* rd, range_t* r)
{
	d->done = 0; d->foundRange = 0; d->lastRange = NULL; d->delayedRange = NULL;
	d->patsrc = rd;
	d->rnd = rd[0].seed;                                 /* patsrc->bufa().seed */
	if (d->nrss == 0) return;
	for (int i = 0; i < d->nrss; i++) drv_set_query(d->rss[i], rd, r);
	memcpy(d->active, d->rss, sizeof(driver_t*) * (size_t)d->nrss); d->nactive = d->nrss;
	d->minCost = 0;
	cost_sort_actives(d);
}

/* addSource, :2098-2111 */
static void cost_add_source(driver_t* d, driver_t* p, range_t* r)
{
	d->lastRange = NULL; d->delayedRange = NULL; d->done = 0;
	if (d->patsrc) drv_set_query(p, d->patsrc, r);
	cost_add_rss(d, p);
	d->active[d->nactive++] = p;
	cost_calc_paired(d);
	d->minCost = 0;
	cost_sort_actives(d);
}

/* clearSources, :2116-2124 */
static void cost_clear_sources(driver_t* d)
{
	for (int i = 0; i < d->nrss; i++) drv_free(d->rss[i]);
	d->nrss = d->nactive = 0; d->paired = 0;
}

/* foundFirstRange, :2311-2362.  Note rss_[i] (not active_[i]) supplies mate1()/fw(). */
static int cost_found_first_range(driver_t* d, range_t* r)
{
	d->foundRange = 1;
	d->lastRange = r;
	if (d->strandFix) {
		int sz = d->nactive;
		for (int i = 1; i < sz; i++) {
			if ((d->rss[i]->mate1 != 0) == (r->mate1 != 0) && d->rss[i]->fw != r->fw) {
				driver_t* p = d->active[i];
				uint16_t minCost = d->minCost > p->minCost ? d->minCost : p->minCost;
				if (minCost > r->cost) break;
				while (!p->done && !p->foundRange) {
					drv_advance(p, ADV_COST_CHANGES);
					if (p->minCost > minCost) break;
				}
				if (p->foundRange) {
					d->delayedRange = drv_range(p);
					size_t tot = (size_t)(d->delayedRange->bot - d->delayedRange->top) + (d->lastRange->bot - d->lastRange->top);
					uint32_t rq = (uint32_t)(rnd_next(&d->rnd) % tot);
					if (rq < d->delayedRange->bot - d->delayedRange->top) {
						range_t* t = d->lastRange; d->lastRange = d->delayedRange; d->delayedRange = t;
					}
					p->foundRange = 0;
				}
				return 1;
			}
		}
	}
	return 0;
}

/* advanceImpl, :2157-2210 */
static void cost_advance(driver_t* d, int until)
{
	d->lastRange = NULL;
	int actSz = d->nactive;
	if (d->delayedRange != NULL) {
		d->lastRange = d->delayedRange; d->delayedRange = NULL; d->foundRange = 1;
		if (d->nactive > 0) d->minCost = d->active[0]->minCost > d->minCost ? d->active[0]->minCost : d->minCost;
		else d->done = 1;
		return;
	}
	if (cost_mate_eliminated(d) || actSz == 0) { d->nactive = 0; d->done = 1; return; }
	driver_t* p = d->active[0];
	uint16_t precost = p->minCost;
	if (!p->foundRange) drv_advance(p, until);
	int needsSort = 0;
	if (p->foundRange) {
		range_t* r = drv_range(p);
		needsSort = cost_found_first_range(d, r);
		p->foundRange = 0;
	}
	if (p->done || precost != p->minCost || needsSort) {
		cost_sort_actives(d);
		if (cost_mate_eliminated(d) || d->nactive == 0) { d->nactive = 0; d->done = (d->delayedRange == NULL); }
	}
}

/* ---- EbwtSeededRangeSourceDriver, ebwt_search_backtrack.h:2935-3141 ---- */
static driver_t* seeded_new(env_t* env, int maq, int qualOrder, const single_spec_t* fact, driver_t* rsSeed, int fw)
{
	driver_t* d = (driver_t*)calloc(1, sizeof(driver_t));
	d->type = DR_SEEDED; d->done = 1; d->env = env; d->maq = maq; d->qualOrder = qualOrder;
	d->rsFact = *fact; d->rsSeed = rsSeed; d->fw = fw; d->mate1 = fact->mate1;
	d->rsFull = cost_new(env, 0);
	return d;
}

/* setQueryImpl, :2963-2978 */
static void seeded_set_query(driver_t* d, const read_t* rd, range_t* partial)
{
	d->done = 0;
	drv_set_query(d->rsSeed, rd, partial);
	d->minCostAdjustment = d->rsSeed->minCostAdjustment > d->rsSeed->minCost ? d->rsSeed->minCostAdjustment : d->rsSeed->minCost;
	d->minCost = d->minCostAdjustment;
	cost_clear_sources(d->rsFull);
	drv_set_query(d->rsFull, rd, partial);
	d->rsFull->minCost = d->minCost;
	d->foundRange = 0;
}

/* advanceImpl, :3011-3103 */
static void seeded_advance(driver_t* d, int until)
{
	driver_t *seed = d->rsSeed, *full = d->rsFull;
	if (seed->done && full->done && !seed->foundRange && !full->foundRange) { d->done = 1; return; }
	if (seed->done && !seed->foundRange) {
		seed->minCost = 0xffff;
		if (full->minCost > d->minCost) { d->minCost = full->minCost; return; }
	}
	if (full->done && !full->foundRange) {
		full->minCost = 0xffff;
		if (seed->minCost > d->minCost) { d->minCost = seed->minCost; return; }
	}
	int doFull = full->minCost <= seed->minCost;
	if (!doFull) {
		if (!seed->foundRange) drv_advance(seed, until);
		if (seed->foundRange) {
			d->seedRange = drv_range(seed);
			seed->foundRange = 0;
			d->minCostAdjustment = d->seedRange->cost;
			driver_t* partial = single_new(d->env, d->maq, d->qualOrder, &d->rsFact);
			partial->minCost = d->seedRange->cost;
			full->minCost = d->seedRange->cost;
			cost_add_source(full, partial, d->seedRange);
			if (full->foundRange) { d->foundRange = 1; full->foundRange = 0; }
		}
		if (seed->minCost > d->minCost) {
			d->minCost = seed->minCost;
			if (!full->done) d->minCost = d->minCost < full->minCost ? d->minCost : full->minCost;
		}
	} else {
		uint16_t oldFullCost = full->minCost;
		if (!full->foundRange) drv_advance(full, until);
		if (full->foundRange) { d->foundRange = 1; full->foundRange = 0; }
		if (full->minCost > oldFullCost) d->minCost = full->minCost < seed->minCost ? full->minCost : seed->minCost;
	}
}

/* ---- virtual dispatch ---- */
static void drv_set_query(driver_t* d, const read_t* rd, range_t* r)
{
	if (d->type == DR_SINGLE) single_set_query(d, rd, r);
	else if (d->type == DR_COST) cost_set_query(d, rd, r);
	else seeded_set_query(d, rd, r);
}
static void drv_advance(driver_t* d, int until)
{
	if (d->type == DR_SINGLE) single_advance(d, until);
	else {
		/* CostAware...::advance (range_source.h:2144-2154), EbwtSeeded...::advance (:2984-2991) */
		if (until < ADV_COST_CHANGES) until = ADV_COST_CHANGES;
		if (d->type == DR_COST) cost_advance(d, until); else seeded_advance(d, until);
	}
}
static range_t* drv_range(driver_t* d)
{
	if (d->type == DR_SINGLE) { d->rs->curRange.fw = d->fw; d->rs->curRange.mate1 = d->mate1; return &d->rs->curRange; }
	if (d->type == DR_COST) return d->lastRange;
	range_t* r = d->rsFull->lastRange; r->fw = d->fw; r->mate1 = d->mate1; return r;
}

/* ------------------------------------------------------------------------------------------
 * Driver trees: one per mode, exactly as the Unpaired*Factory::create() bodies build them
 * ---------------------------------------------------------------------------------------- */
#define B PIN_TO_BEGINNING
#define L PIN_TO_LEN
#define H PIN_TO_HI_HALF_EDGE
#define S PIN_TO_SEED_EDGE

static int g_mate1 = 1;     /* mate the specs being built are for (tree construction only) */
static single_spec_t spec(const bto_index* ebwt, int fw, uint32_t qualLim, int reportExacts, int hh, int partial,
                          int seed, uint32_t seedLen, int nudgeLeft, int r0, int r1, int r2, int r3, int* btCnt)
{
	single_spec_t sp;
	sp.ebwt = ebwt; sp.fw = fw; sp.qualLim = qualLim; sp.reportExacts = reportExacts; sp.halfAndHalf = hh;
	sp.partial = partial; sp.seed = seed; sp.seedLen = seedLen; sp.nudgeLeft = nudgeLeft;
	sp.rev[0] = r0; sp.rev[1] = r1; sp.rev[2] = r2; sp.rev[3] = r3; sp.btCnt = btCnt;
	sp.mate1 = g_mate1;
	return sp;
}

/* the drivers of one (mate, strand) combination, in the order the factories push them */
static void add_block(driver_t* top, env_t* env, const bto_index* fwI, const bto_index* bwI, const bt_policy* pol,
                      int* btCnt, int fw, int paired)
{
	int maq = pol->maq_round, qo = 1 /* qualOrder = !better */;
	single_spec_t sp, fs;
	/* fw read: mirror index first; rc read: text index first */
	const bto_index* i1 = fw ? bwI : fwI;
	const bto_index* i2 = fw ? fwI : bwI;
#define ADD1(...) do { sp = spec(__VA_ARGS__); cost_add_rss(top, single_new(env, maq, qo, &sp)); } while (0)
#define ADDSEED(fact_args, gen_args) do { \
		fs = spec fact_args; sp = spec gen_args; \
		cost_add_rss(top, seeded_new(env, maq, qo, &fs, single_new(env, maq, qo, &sp), fw)); } while (0)
	if (pol->mode == BT_MODE_V) {
		if (pol->mms == 0) {                       /* aligner_0mm.h:69-115, 244-300 */
			ADD1(fwI, fw, OFF_MASK, 1, 0, 0, 0, 0, 1, L, L, L, L, NULL);
		} else if (pol->mms == 1) {                /* aligner_1mm.h:73-152, 286-415: nudgeLeft is true on the text index */
			/* nudgeLeft: "true for Fw index" in the unpaired factory; the paired one (aligner_1mm.h:295-408)
			 * passes true for the first and false for the second driver of every block */
			ADD1(i1, fw, OFF_MASK, 1, 0, 0, 0, 0, paired ? 1 : i1 == fwI, H, L, L, L, NULL);
			ADD1(i2, fw, OFF_MASK, 0, 0, 0, 0, 0, paired ? 0 : i2 == fwI, H, L, L, L, NULL);
		} else {                                   /* aligner_23mm.h:73-236, 358-606: nudgeLeft alternates */
			int two = (pol->mms == 2);
			int r2 = two ? L : H;
			ADD1(i1, fw, OFF_MASK, 1, 0, 0, 0, 0, 1, H, H, r2, L, NULL);
			ADD1(i2, fw, OFF_MASK, 0, 0, 0, 0, 0, 0, H, H, r2, L, NULL);
			ADD1(i1, fw, OFF_MASK, 0, 2, 0, 0, 0, 1, B, H, r2, L, NULL);
			/* the 3-mismatch half-and-half driver: rev1Off is PIN_TO_HI_HALF_EDGE in the unpaired factory and
			 * for mate 1's rc block of the paired one, PIN_TO_BEGINNING in the paired factory's other three
			 * blocks (aligner_23mm.h:403-411, 469-477, 534-542, 598-606) */
			if (!two) ADD1(i2, fw, OFF_MASK, 0, 3, 0, 0, 0, 0, B, (paired && !(g_mate1 && !fw)) ? B : H, H, L, NULL);
		}
	} else {                                           /* aligner_seed_mm.h:82-516, 705-1290 */
		uint32_t q = (uint32_t)pol->qual_thresh, sl = (uint32_t)pol->seed_len;
		int n = pol->mms;
		int* bc = n >= 2 ? btCnt : NULL;
		if (n == 0) {
			ADD1(i1, fw, q, 1, 0, 0, 0, sl, 1, S, S, S, S, NULL);
		} else {
			int a1 = n >= 2 ? H : S, a2 = n >= 3 ? H : S;
			ADD1(i1, fw, q, 1, 0, 0, 0, sl, 1, H, a1, a2, S, bc);
			ADDSEED((i1, fw, q, 1, 0, 0, 0, sl, 1, S, S, S, S, bc), (i2, fw, q, 0, 0, 1, 1, sl, 0, H, a1, a2, S, bc));
			if (n >= 3) ADDSEED((i1, fw, q, 1, 0, 0, 0, sl, 1, S, S, S, S, bc), (i2, fw, q, 0, 3, 1, 1, sl, 0, B, H, H, S, bc));
			if (n >= 2) ADD1(i1, fw, q, 0, 2, 0, 0, sl, 1, B, H, a2, S, bc);
		}
	}
#undef ADD1
#undef ADDSEED
}

static driver_t* build_tree(env_t* env, const bto_index* fwI, const bto_index* bwI, const bt_policy* pol, int* btCnt)
{
	driver_t* top = cost_new(env, 1 /* strandFix default (ebwt_search.cpp:227) */);
	g_mate1 = 1;
	if (!pol->nofw) add_block(top, env, fwI, bwI, pol, btCnt, 1, 0);
	if (!pol->norc) add_block(top, env, fwI, bwI, pol, btCnt, 0, 0);
	return top;
}

/* Paired*AlignerFactory::create() with v1_ == false (--best): every driver of both mates in one
 * cost-aware driver; -v: 1Fw 1Rc 2Fw 2Rc (aligner_0mm.h:320-327, aligner_1mm.h:286-415,
 * aligner_23mm.h:358-606), -n: 1Fw 2Fw 1Rc 2Rc (aligner_seed_mm.h:705-1290) */
static driver_t* build_tree_paired(env_t* env, const bto_index* fwI, const bto_index* bwI, const bt_policy* pol, int* btCnt)
{
	driver_t* top = cost_new(env, 1);
	int do1Fw = 1, do1Rc = 1, do2Fw = 1, do2Rc = 1;
	if (pol->nofw) { if (pol->mate1_fw) do1Fw = 0; else do1Rc = 0; if (pol->mate2_fw) do2Fw = 0; else do2Rc = 0; }
	if (pol->norc) { if (pol->mate1_fw) do1Rc = 0; else do1Fw = 0; if (pol->mate2_fw) do2Rc = 0; else do2Fw = 0; }
	int order_v[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}}, order_n[4][2] = {{1, 1}, {0, 1}, {1, 0}, {0, 0}};
	for (int k = 0; k < 4; k++) {
		int mate1 = pol->mode == BT_MODE_V ? order_v[k][0] : order_n[k][0];
		int fw = pol->mode == BT_MODE_V ? order_v[k][1] : order_n[k][1];
		int doit = mate1 ? (fw ? do1Fw : do1Rc) : (fw ? do2Fw : do2Rc);
		if (!doit) continue;
		g_mate1 = mate1;
		add_block(top, env, fwI, bwI, pol, btCnt, fw, 1);
	}
	g_mate1 = 1;
	cost_calc_paired(top);
	return top;
}
#undef B
#undef L
#undef H
#undef S

/* ------------------------------------------------------------------------------------------
 * Sinks: NGood hit.h:969-985, NBestFirstStrat hit.h:1070-1129, All hit.h:1201-1209
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t n, max, hitsForThisRead;
	int strata, all, bestStratum;
	bto_hit* hits; int cap, stored, dropped;
} sink_t;

static int sink_report(sink_t* s, const bto_hit* h, int stratum)
{
	s->hitsForThisRead++;
	if (s->strata && stratum < s->bestStratum) s->bestStratum = stratum;
	if (s->hitsForThisRead > s->max) return 1;
	if (s->stored < s->cap) s->hits[s->stored++] = *h; else s->dropped = 1;
	if (s->all && !s->strata) return 0;
	if (s->hitsForThisRead == s->n && (s->max == 0xffffffffu || s->max < s->n)) return 1;
	return 0;
}
static int sink_irrelevant_cost(const sink_t* s, uint16_t cost)
{
	if (!s->strata) return 0;
	if (s->hitsForThisRead) return ((int)cost >> 14) > s->bestStratum;
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * RowChaser + RangeChaser (no range cache: ebwt_search.cpp passes NULL caches), UnpairedAlignerV2
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	const bto_index* ebwt; uint32_t qlen, top, bot, irow, row;
	uint32_t offTidx, offToff, tlen;     /* off_; offTidx == OFF_MASK: not found */
	int done;
	/* RowChaser */
	int cDone; uint32_t cRow, cJumps, cOff;
	uint64_t* probes;
} chaser_t;

/* RowChaser::setRow, row_chaser.h:69-95 */
static void rowchaser_set_row(chaser_t* c, uint32_t row, env_t* env)
{
	const bto_index* ix = c->ebwt;
	c->cRow = row;
	if (row == ix->zOff) { c->cOff = 0; c->cDone = 1; return; }
	if ((row & ix->offMask) == row) { c->cOff = ix->offs[row >> ix->offRate]; c->cDone = 1; if (env->cnt) env->cnt->offs++; return; }
	c->cDone = 0; c->cJumps = 0; c->cOff = OFF_MASK;
}
/* RowChaser::advance, row_chaser.h:99-123 (one call runs the walk to its end) */
static void rowchaser_advance(chaser_t* c, env_t* env)
{
	const bto_index* ix = c->ebwt;
	while (!c->cDone) {
		uint32_t lf[4];
		bto_rank4(ix, c->cRow, lf);
		c->cRow = lf[bto_rowL(ix, c->cRow)];
		c->cJumps++;
		if (env->cnt) env->cnt->chase++;
		if (c->cRow == ix->zOff) { c->cOff = c->cJumps; c->cDone = 1; }
		else if ((c->cRow & ix->offMask) == c->cRow) { c->cOff = ix->offs[c->cRow >> ix->offRate] + c->cJumps; c->cDone = 1; if (env->cnt) env->cnt->offs++; }
	}
}
/* RowChaser::off, row_chaser.h:146-155 */
static void rowchaser_off(chaser_t* c, uint32_t* tidx, uint32_t* toff)
{
	*toff = OFF_MASK;
	if (!bto_joined_to_text_cnt(c->ebwt, c->qlen, c->cOff, tidx, toff, &c->tlen, c->probes)) *tidx = OFF_MASK;
}

/* RangeChaser::setRow, range_chaser.h:52-121 */
static void chaser_set_row(chaser_t* c, uint32_t row, env_t* env)
{
	c->row = row;
	for (;;) {
		rowchaser_set_row(c, c->row, env);
		if (c->cDone) {
			rowchaser_off(c, &c->offTidx, &c->offToff);
			if (c->offTidx != OFF_MASK) return;
		} else break;
		c->row++;
		if (c->row == c->bot) c->row = c->top;
		if (c->row == c->irow) { c->done = 1; return; }
	}
}
/* RangeChaser::setTopBot, range_chaser.h:126-166 */
static void chaser_set_top_bot(chaser_t* c, uint32_t top, uint32_t bot, uint32_t qlen, uint32_t* rnd,
                               const bto_index* ebwt, env_t* env)
{
	c->ebwt = ebwt; c->qlen = qlen; c->top = top; c->bot = bot;
	uint32_t spread = bot - top;
	c->irow = top + (rnd_next(rnd) % spread);
	c->done = 0;
	c->offTidx = OFF_MASK;
	chaser_set_row(c, c->irow, env);
}
/* RangeChaser::advance, range_chaser.h:171-209 */
static void chaser_advance(chaser_t* c, env_t* env)
{
	c->offTidx = OFF_MASK;
	if (c->cDone) {
		c->row++;
		if (c->row == c->bot) c->row = c->top;
		if (c->row == c->irow) { c->done = 1; return; }
		chaser_set_row(c, c->row, env);
	} else {
		rowchaser_advance(c, env);
		if (c->cDone) rowchaser_off(c, &c->offTidx, &c->offToff);
	}
}

/* UnpairedAlignerV2::report (aligner.h:467-497) + EbwtSearchParams::reportHit (ebwt.h:1288-1405) */
static int al_report_ex(sink_t* sink, const range_t* ra, uint32_t tidx, uint32_t toff, uint32_t alen,
                        int ebwtFw, int mate, uint32_t oms)
{
	bto_hit h;
	memset(&h, 0, sizeof(h));
	h.tidx = tidx; h.toff = toff; h.oms = oms; h.mate = (uint16_t)mate;
	h.cost = ra->cost; h.stratum = (uint8_t)ra->stratum; h.fw = (uint8_t)ra->fw;
	uint32_t n = ra->numMms > BTO_MAXMM ? BTO_MAXMM : ra->numMms;
	h.nmm = (uint16_t)n;
	int flip = (ebwtFw != 0) != (ra->fw != 0);
	for (uint32_t i = 0; i < n; i++) {
		uint32_t pos = flip ? alen - ra->mms[i] - 1 : ra->mms[i];
		uint8_t rc = ra->refcs[i];
		uint32_t code = rc == 'A' ? 0 : rc == 'C' ? 1 : rc == 'G' ? 2 : 3;
		h.mm[i] = (uint16_t)(pos | (code << 12));
	}
	for (uint32_t i = 1; i < n; i++) {          /* Hit::mms is a bitset: order by position */
		uint16_t v = h.mm[i]; int j = (int)i - 1;
		while (j >= 0 && BT_MM_POS(h.mm[j]) > BT_MM_POS(v)) { h.mm[j + 1] = h.mm[j]; j--; }
		h.mm[j + 1] = v;
	}
	return sink_report(sink, &h, (int)ra->stratum);
}
static int al_report(sink_t* sink, const range_t* ra, uint32_t tidx, uint32_t toff, uint32_t alen)
{
	return al_report_ex(sink, ra, tidx, toff, alen, ra->ebwt->fw != 0, 0, ra->bot - ra->top - 1);
}

int bto_align_read_best(const bto_index* ixFw, const bto_index* ixBw, const bt_policy* pol,
                        const uint8_t* seq, const uint8_t* qual, int len, uint32_t seed,
                        bto_hit* hits, int cap, uint32_t* n_hits_total, uint32_t* status,
                        bt_op_counts* counts)
{
	if (len < 0 || len > BTO_MAXLEN) return -BT_ERR_ARG;
	if (pol->mode == BT_MODE_V ? (pol->mms < 0 || pol->mms > 3) : (pol->mms < 0 || pol->mms > 3)) return -BT_ERR_ARG;
	if ((pol->mode == BT_MODE_N || pol->mms > 0) && !ixBw) return -BT_ERR_ARG;
	sink_t sink;
	memset(&sink, 0, sizeof(sink));
	/* createSinkFactory, ebwt_search.cpp:992-1020 */
	sink.strata = pol->strata;
	sink.all = pol->all_hits;
	sink.n = pol->all_hits ? (pol->strata ? 0xffffffffu / 2 : 0xffffffffu) : pol->khits;
	sink.max = pol->mhits;
	sink.bestStratum = 999;
	sink.hits = hits; sink.cap = cap;
	uint32_t st = 0;
	int budget = BRANCH_BUDGET;
	env_t env = { &budget, counts };
	/* UnpairedAlignerV2::setQuery, aligner.h:434-462 */
	if (len < 4) {
		st |= BT_ST_SKIPPED;
		if (n_hits_total) *n_hits_total = 0;
		if (status) *status = st;
		return 0;
	}
	read_t* rd = (read_t*)calloc(2, sizeof(read_t));
	read_init(rd, seq, qual, (uint32_t)len, seed);
	int btCntStore = pol->max_bts;
	int* btCnt = (pol->mode == BT_MODE_N && pol->mms >= 2) ? &btCntStore : NULL;
	driver_t* driver = build_tree(&env, ixFw, ixBw, pol, btCnt);
	uint32_t alRnd = seed;                       /* Aligner::rand_ (aligner.h:65) */
	chaser_t ch; memset(&ch, 0, sizeof(ch));
	ch.probes = counts ? &counts->rstarts : NULL;
	drv_set_query(driver, rd, NULL);
	int done = driver->done;
	if (btCnt) *btCnt = pol->max_bts;
	int chase = 0;
	/* UnpairedAlignerV2::advance until done, aligner.h:503-567 */
	while (!done) {
		if (chase) {
			if (ch.offTidx == OFF_MASK && !ch.done) { chaser_advance(&ch, &env); continue; }
			if (ch.offTidx != OFF_MASK) {
				done = al_report(&sink, drv_range(driver), ch.offTidx, ch.offToff, (uint32_t)len);
				ch.offTidx = OFF_MASK;
			} else {
				chase = 0;
				driver->foundRange = 0;
				done = driver->done;
			}
		}
		if (!done && !chase) {
			if (driver->foundRange) {
				const range_t* ra = drv_range(driver);
				chaser_set_top_bot(&ch, ra->top, ra->bot, (uint32_t)len, &alRnd, ra->ebwt, &env);
				if (ch.offTidx != OFF_MASK) {
					done = al_report(&sink, ra, ch.offTidx, ch.offToff, (uint32_t)len);
					ch.offTidx = OFF_MASK;
				}
				if (!ch.done && !sink_irrelevant_cost(&sink, ra->cost)) chase = 1;
				else driver->foundRange = 0;
			} else {
				done = sink_irrelevant_cost(&sink, driver->minCost);
				if (!done) drv_advance(driver, ADV_COST_CHANGES);
			}
			if (driver->done && !driver->foundRange && !chase) done = 1;
		}
	}
	if (budget < 0) st |= BT_ST_OVERFLOW;
	drv_free(driver);
	free(rd);
	if (sink.dropped && sink.hitsForThisRead <= sink.max) st |= BT_ST_HITCAP;
	if (n_hits_total) *n_hits_total = sink.hitsForThisRead;
	if (status) *status = st;
	/* finishRead (hit.h:741-786) + NBestFirstStrat::finishReadImpl (hit.h:1098-1110): with --strata
	 * every buffered hit's oms becomes (#buffered - 1) */
	if (sink.strata) for (int i = 0; i < sink.stored; i++) sink.hits[i].oms = (uint32_t)sink.stored - 1;
	if (sink.hitsForThisRead > sink.max) {
		/* maxed: nothing is reported unless -M; the buffered hits (first _max) are what
		 * reportMaxed samples from (hit.cpp:16-68) */
		return pol->sample_max ? sink.stored : 0;
	}
	int n = sink.stored;
	if ((uint32_t)n > sink.n) n = (int)sink.n;
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Paired-end: BitPairReference (reference.h:35-120, getStretch :479), RefAligner::find
 * (ref_aligner.h:63-101; the naiveFind of each concrete aligner is the specification its
 * anchor64Find is asserted against: :182-262, 513-601, 2561-2752), PairedBWAlignerV2
 * (aligner.h:1483-2051)
 * ---------------------------------------------------------------------------------------- */
struct bto_refs {
	uint32_t n;
	uint8_t** seq;           /* [tidx][plen]: 0..3, 4 = N / gap */
	uint32_t* approxLen;     /* BitPairReference::approxLen: up to the end of the last unambiguous stretch */
	uint32_t* plen;
};

/* The reference's BitPairReference reads <base>.3.ebwt / .4.ebwt; the oracle derives the same
 * sequences independently from the index proper: the joined text (Ebwt::restore) cut at the
 * fragment table rstarts[] (joined offset, tidx, offset within the reference). */
bto_refs* bto_refs_build(const bto_index* ix)
{
	bto_refs* r = (bto_refs*)calloc(1, sizeof(bto_refs));
	r->n = ix->nPat;
	r->seq = (uint8_t**)calloc(ix->nPat, sizeof(uint8_t*));
	r->approxLen = (uint32_t*)calloc(ix->nPat, 4);
	r->plen = (uint32_t*)calloc(ix->nPat, 4);
	uint8_t* text = (uint8_t*)malloc(ix->len ? ix->len : 1);
	bto_restore_text(ix, text);
	for (uint32_t t = 0; t < ix->nPat; t++) {
		r->plen[t] = ix->plen[t];
		r->seq[t] = (uint8_t*)malloc(ix->plen[t] + 16);
		memset(r->seq[t], 4, ix->plen[t] + 16);
	}
	for (uint32_t f = 0; f < ix->nFrag; f++) {
		uint32_t joff = ix->rstarts[f * 3], tidx = ix->rstarts[f * 3 + 1], foff = ix->rstarts[f * 3 + 2];
		uint32_t jend = (f + 1 < ix->nFrag) ? ix->rstarts[(f + 1) * 3] : ix->len;
		uint32_t flen = jend - joff;
		memcpy(r->seq[tidx] + foff, text + joff, flen);
		if (foff + flen > r->approxLen[tidx]) r->approxLen[tidx] = foff + flen;
	}
	free(text);
	return r;
}
void bto_refs_free(bto_refs* r)
{
	if (!r) return;
	for (uint32_t t = 0; t < r->n; t++) free(r->seq[t]);
	free(r->seq); free(r->approxLen); free(r->plen); free(r);
}

/* the set of (upstream, downstream) coordinate pairs already reported for one pair orientation
 * (TSetPairs; aligner.h:2049) */
typedef struct { uint64_t* a; int n, cap; } pairset_t;
static int pairset_has(const pairset_t* p, uint64_t first, uint64_t second)
{
	for (int i = 0; i < p->n; i++) if (p->a[2 * i] == first && p->a[2 * i + 1] == second) return 1;
	return 0;
}
static void pairset_add(pairset_t* p, uint64_t first, uint64_t second)
{
	if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 16; p->a = (uint64_t*)realloc(p->a, 16 * (size_t)p->cap); }
	p->a[2 * p->n] = first; p->a[2 * p->n + 1] = second; p->n++;
}

/* RefAligner::find with numToFind = 1: candidate leftmost positions radiate out from the middle of
 * [begin, end - qlen]; the first one that (a) touches no reference N, (b) has at most `k`
 * mismatches in the seed (the first min(qlen, seedLen) bases from the 5' end; the whole read in
 * -v mode), (c) for the seeded aligners keeps the summed penalty of all its mismatches within
 * qualMax, and (d) is not in `pairs` yet, is returned.  Returns 1 and fills *r / *result. */
static int ref_find_one(const bt_policy* pol, const uint8_t* ref, const uint8_t* qry, const uint8_t* quals, uint32_t qlen,
                        uint32_t tidx, uint32_t begin, uint32_t end, int seedOnLeft, pairset_t* pairs, uint32_t aoff,
                        range_t* r, uint32_t* result)
{
	const int seeded = pol->mode == BT_MODE_N;
	const uint32_t k = (uint32_t)pol->mms;
	const uint32_t slen = seeded ? (qlen < (uint32_t)pol->seed_len ? qlen : (uint32_t)pol->seed_len) : qlen;
	const uint32_t qualMax = seeded ? (uint32_t)pol->qual_thresh : 0xffffffffu;
	const uint32_t lim = end - qlen - begin;
	const uint32_t halfway = begin + (lim >> 1);
	int hi = 0;
	for (uint32_t i = 1; i <= lim + 1; i++) {
		uint32_t ri = hi ? halfway + (i >> 1) : halfway - (i >> 1);
		hi = !hi;
		int match = 1;
		uint32_t mms = 0, seedMms = 0, ham = 0;
		r->n = 0;
		for (uint32_t j = 0; j < qlen; j++) {
			int rc = ref[ri + j];
			if (rc & 4) { match = 0; break; }
			if (qry[j] != rc) {
				const int inSeed = seedOnLeft ? (j < slen) : (j >= qlen - slen);
				if (inSeed && ++seedMms > k) { match = 0; break; }
				if (seeded) {
					ham += mm_penalty(pol->maq_round, phred_of(quals[j]));
					if (ham > qualMax) { match = 0; break; }
				}
				r->mms[mms] = j; r->refcs[mms] = (uint8_t)"ACGT"[rc]; mms++;
			}
		}
		if (!match) continue;
		if (pairs) {
			uint64_t a = ((uint64_t)tidx << 32) | ri, b = ((uint64_t)tidx << 32) | aoff;
			uint64_t first = ri < aoff ? a : b, second = ri < aoff ? b : a;
			if (pairset_has(pairs, first, second)) continue;
			pairset_add(pairs, first, second);
		}
		r->n = mms; r->numMms = mms; r->stratum = seedMms; r->cost = 0; r->ebwt = NULL;
		*result = ri;
		return 1;
	}
	return 0;
}

int bto_align_pair_best(const bto_index* ixFw, const bto_index* ixBw, const bto_refs* refs, const bt_policy* pol,
                        const uint8_t* seq1, const uint8_t* qual1, int len1, uint32_t seed1,
                        const uint8_t* seq2, const uint8_t* qual2, int len2, uint32_t seed2,
                        bto_hit* hits, int cap, uint32_t* n_hits_total, uint32_t* status, bt_op_counts* counts)
{
	if (len1 < 0 || len1 > BTO_MAXLEN || len2 < 0 || len2 > BTO_MAXLEN || !refs) return -BT_ERR_ARG;
	if ((pol->mode == BT_MODE_N || pol->mms > 0) && !ixBw) return -BT_ERR_ARG;
	sink_t sink;
	memset(&sink, 0, sizeof(sink));
	/* createMult(2) (hit.h:1012-1016, 1155-1159, 1246-1249): mates count separately */
	sink.strata = pol->strata; sink.all = pol->all_hits;
	sink.n = pol->all_hits ? (pol->strata ? (0xffffffffu / 2) * 2u : 0xffffffffu) : pol->khits * 2u;
	sink.max = pol->mhits == 0xffffffffu ? 0xffffffffu : pol->mhits * 2u;
	sink.bestStratum = 999;
	sink.hits = hits; sink.cap = cap;
	uint32_t st = 0;
	int budget = BRANCH_BUDGET;
	env_t env = { &budget, counts };
	if (len1 < 4 || len2 < 4) {                 /* aligner.h:1579-1588 */
		if (n_hits_total) *n_hits_total = 0;
		if (status) *status = BT_ST_SKIPPED;
		return 0;
	}
	read_t* rd = (read_t*)calloc(2, sizeof(read_t));
	read_init(&rd[0], seq1, qual1, (uint32_t)len1, seed1);
	read_init(&rd[1], seq2, qual2, (uint32_t)len2, seed2);
	int btCntStore = pol->max_bts;
	int* btCnt = (pol->mode == BT_MODE_N && pol->mms >= 2) ? &btCntStore : NULL;
	driver_t* driver = build_tree_paired(&env, ixFw, ixBw, pol, btCnt);
	uint32_t alRnd = seed1;                       /* Aligner::rand_.init(bufa_->seed) */
	chaser_t ch; memset(&ch, 0, sizeof(ch));
	ch.probes = counts ? &counts->rstarts : NULL;
	drv_set_query(driver, rd, NULL);
	if (btCnt) *btCnt = pol->max_bts;
	const int fw1 = pol->mate1_fw, fw2 = pol->mate2_fw;
	pairset_t pairs_fw = {0, 0, 0}, pairs_rc = {0, 0, 0};
	range_t* found = (range_t*)malloc(sizeof(range_t));
	uint32_t mixedAttempts = 0;
	int done = 0, chase = 0, donePe = 0;
	while (!done) {
		if (chase) {
			if (ch.offTidx == OFF_MASK && !ch.done) { chaser_advance(&ch, &env); continue; }
			if (ch.offTidx != OFF_MASK) {
				/* resolveOutstanding (aligner.h:1849-1871) -> resolveOutstandingInRef (:1883-1997) */
				const range_t* range = drv_range(driver);
				const uint32_t tidx = ch.offTidx, toff = ch.offToff, tlen = ixFw->plen[ch.offTidx];
				(void)tlen;
				if (!donePe) {
					int ret = 0;
					const int pairFw = range->mate1 ? (range->fw == fw1) : (range->fw == fw2);
					const int matchRight = pairFw ? range->mate1 : !range->mate1;
					int fw = range->mate1 ? fw2 : fw1;
					if (!pairFw) fw = !fw;
					const read_t* om = &rd[range->mate1 ? 1 : 0];             /* the outstanding mate */
					const uint8_t* oseq = fw ? om->pat[1][1] : om->pat[0][1];   /* patFw : patRc */
					const uint8_t* oqual = fw ? om->qual[0] : om->qual[1];
					const uint32_t qlen = om->len, alen = range->mate1 ? rd[0].len : rd[1].len;
					const int minins = pol->min_ins, maxins = pol->max_ins;
					if ((uint32_t)maxins > (qlen > alen ? qlen : alen)) {
						uint32_t begin, end;
						const uint32_t insDiff = (uint32_t)(maxins - minins);
						const int contain = pol->allow_contain;
						if (matchRight) {
							end = toff + (uint32_t)maxins;
							begin = toff + (contain ? 0 : 1);
							if (!contain && qlen < alen) begin += alen - qlen;
							if (end > insDiff + qlen) { uint32_t b2 = end - insDiff - qlen; if (b2 > begin) begin = b2; }
							if (refs->approxLen[tidx] < end) end = refs->approxLen[tidx];
							if (refs->approxLen[tidx] < begin) begin = refs->approxLen[tidx];
						} else {
							begin = (toff + alen < (uint32_t)maxins) ? 0 : toff + alen - (uint32_t)maxins;
							const uint32_t mi = alen < qlen ? alen : qlen;
							if (contain) end = toff + alen - 1;
							else {
								end = toff + mi - 1;
								const uint32_t e2 = toff + alen - (uint32_t)minins + qlen - 1;
								if (e2 < end) end = e2;
								if (toff + alen + qlen < (uint32_t)minins + 1) end = 0;
							}
						}
						if (end >= begin && end - begin >= qlen) {   /* aligner.h:1964 */
							uint32_t result = 0;
							if (ref_find_one(pol, refs->seq[tidx], oseq, oqual, qlen, tidx, begin, end, fw,
							                 pairFw ? &pairs_fw : &pairs_rc, toff, found, &result)) {
								range_t* r = found;
								r->fw = fw; r->cost = (uint16_t)(r->cost | (r->stratum << 14));
								r->mate1 = !range->mate1; r->top = range->top; r->bot = range->bot;
								const int ebwtLFw = matchRight ? (range->ebwt->fw != 0) : 1;
								const int ebwtRFw = matchRight ? 1 : (range->ebwt->fw != 0);
								const range_t* rL = matchRight ? range : r;
								const range_t* rR = matchRight ? r : range;
								const uint32_t up = matchRight ? toff : result, dn = matchRight ? result : toff;
								/* report (aligner.h:1720-1788): upstream mate first */
								const uint32_t oms = (rL->bot - rL->top < rR->bot - rR->top ? rL->bot - rL->top : rR->bot - rR->top) - 1;
								const uint32_t lenL = pairFw ? rd[0].len : rd[1].len, lenR = pairFw ? rd[1].len : rd[0].len;
								ret = al_report_ex(&sink, rL, tidx, up, lenL, ebwtLFw, pairFw ? 1 : 2, oms);
								if (!ret) ret = al_report_ex(&sink, rR, tidx, dn, lenR, ebwtRFw, pairFw ? 2 : 1, oms);
							}
						}
					}
					if (++mixedAttempts > (uint32_t)pol->pair_tries || ret) donePe = 1;
					done = donePe;
				}
				ch.offTidx = OFF_MASK;
			} else {
				chase = 0;
				done = driver->done;
			}
		}
		if (!done && !chase) {
			if (!driver->done) {
				if (!donePe) {
					donePe = sink_irrelevant_cost(&sink, driver->minCost);
					if (donePe) done = 1;
				}
				if (!done) drv_advance(driver, ADV_COST_CHANGES);
				if (driver->foundRange) {
					chase = 1;
					driver->foundRange = 0;
					const range_t* r = drv_range(driver);
					chaser_set_top_bot(&ch, r->top, r->bot, r->mate1 ? rd[0].len : rd[1].len, &alRnd, r->ebwt, &env);
				}
			} else done = 1;
		}
	}
	if (budget < 0) st |= BT_ST_OVERFLOW;
	drv_free(driver);
	free(rd); free(found); free(pairs_fw.a); free(pairs_rc.a);
	if (sink.dropped && sink.hitsForThisRead <= sink.max) st |= BT_ST_HITCAP;
	if (n_hits_total) *n_hits_total = sink.hitsForThisRead;
	if (status) *status = st;
	if (sink.strata) for (int i = 0; i < sink.stored; i++) sink.hits[i].oms = (uint32_t)sink.stored / 2u - 1u;
	if (sink.hitsForThisRead > sink.max) return pol->sample_max ? sink.stored : 0;
	int n = sink.stored;
	if ((uint32_t)n > sink.n) n = (int)sink.n;
	return n;
}


/* ==========================================================================================
 * PairedBWAlignerV1 (aligner.h:606-1480): paired-end WITHOUT --best -- the reference's default
 * ("useV1", ebwt_search.cpp:232,776).  The same four per-mate, per-strand drivers the factories
 * build for V2 (aligner_0mm.h:178-243, aligner_1mm.h:286-432, aligner_23mm.h:358-628,
 * aligner_seed_mm.h:705-1330), but each behind its own cost-aware driver and driven separately:
 * first the pairing in which mate 1 is on its own strand ("fw": L = mate 1, R = mate 2), then the
 * other one (L = mate 2, R = mate 1).  dontReconcileMates is true by default (ebwt_search.cpp:219),
 * so every offset found for a range of one mate goes straight to resolveOutstandingInRef
 * (aligner.h:951-1086): the other mate is looked for in the 2-bit reference.
 * ======================================================================================== */
typedef struct {
	driver_t* drL; driver_t* drR;                /* NULL = StubRangeSourceDriver (range_source.h:1891) */
	int chaseL, chaseR, delayedL, delayedR;
	uint32_t offsLsz, offsRsz;
} v1_orient_t;

#define V1_DONE(d)  ((d) == NULL || (d)->done)
#define V1_FOUND(d) ((d) != NULL && (d)->foundRange)

static driver_t* v1_block(env_t* env, const bto_index* fwI, const bto_index* bwI, const bt_policy* pol, int* btCnt,
                          int mate1, int fw)
{
	g_mate1 = mate1;
	driver_t* d;
	if (pol->mode == BT_MODE_V && pol->mms == 0) {
		/* aligner_0mm.h:178-243: the bare EbwtRangeSourceDriver, no cost-aware wrapper */
		single_spec_t sp = spec(fwI, fw, OFF_MASK, 1, 0, 0, 0, 0, 1, PIN_TO_LEN, PIN_TO_LEN, PIN_TO_LEN, PIN_TO_LEN, NULL);
		d = single_new(env, pol->maq_round, 1, &sp);
	} else {
		d = cost_new(env, 1);                    /* mixedMode = false: no cost_calc_paired */
		add_block(d, env, fwI, bwI, pol, btCnt, fw, 1);
	}
	g_mate1 = 1;
	return d;
}

int bto_align_pair_v1(const bto_index* ixFw, const bto_index* ixBw, const bto_refs* refs, const bt_policy* pol,
                      const uint8_t* seq1, const uint8_t* qual1, int len1, uint32_t seed1,
                      const uint8_t* seq2, const uint8_t* qual2, int len2, uint32_t seed2,
                      bto_hit* hits, int cap, uint32_t* n_hits_total, uint32_t* status, bt_op_counts* counts)
{
	if (len1 < 0 || len1 > BTO_MAXLEN || len2 < 0 || len2 > BTO_MAXLEN || !refs) return -BT_ERR_ARG;
	if ((pol->mode == BT_MODE_N || pol->mms > 0) && !ixBw) return -BT_ERR_ARG;
	sink_t sink;
	memset(&sink, 0, sizeof(sink));
	/* createMult(2): as for V2 */
	sink.strata = pol->strata; sink.all = pol->all_hits;
	sink.n = pol->all_hits ? (pol->strata ? (0xffffffffu / 2) * 2u : 0xffffffffu) : pol->khits * 2u;
	sink.max = pol->mhits == 0xffffffffu ? 0xffffffffu : pol->mhits * 2u;
	sink.bestStratum = 999;
	sink.hits = hits; sink.cap = cap;
	uint32_t st = 0;
	int budget = BRANCH_BUDGET;
	env_t env = { &budget, counts };
	if (len1 < 4 || len2 < 4) {                 /* aligner.h:733-742 */
		if (n_hits_total) *n_hits_total = 0;
		if (status) *status = BT_ST_SKIPPED;
		return 0;
	}
	read_t* rd = (read_t*)calloc(2, sizeof(read_t));
	read_init(&rd[0], seq1, qual1, (uint32_t)len1, seed1);
	read_init(&rd[1], seq2, qual2, (uint32_t)len2, seed2);
	int btCntStore = pol->max_bts;
	int* btCnt = (pol->mode == BT_MODE_N && pol->mms >= 2) ? &btCntStore : NULL;
	const int fw1 = pol->mate1_fw, fw2 = pol->mate2_fw;
	int do1Fw = 1, do1Rc = 1, do2Fw = 1, do2Rc = 1;
	if (pol->nofw) { if (fw1) do1Fw = 0; else do1Rc = 0; if (fw2) do2Fw = 0; else do2Rc = 0; }
	if (pol->norc) { if (fw1) do1Rc = 0; else do1Fw = 0; if (fw2) do2Rc = 0; else do2Fw = 0; }
	driver_t* d1Fw = do1Fw ? v1_block(&env, ixFw, ixBw, pol, btCnt, 1, 1) : NULL;
	driver_t* d1Rc = do1Rc ? v1_block(&env, ixFw, ixBw, pol, btCnt, 1, 0) : NULL;
	driver_t* d2Fw = do2Fw ? v1_block(&env, ixFw, ixBw, pol, btCnt, 0, 1) : NULL;
	driver_t* d2Rc = do2Rc ? v1_block(&env, ixFw, ixBw, pol, btCnt, 0, 0) : NULL;
	driver_t* all[4] = { d1Fw, d1Rc, d2Fw, d2Rc };
	for (int i = 0; i < 4; i++) if (all[i]) drv_set_query(all[i], rd, NULL);     /* aligner.h:743-746 */
	if (btCnt) *btCnt = pol->max_bts;
	uint32_t alRnd = seed1;                       /* Aligner::rand_.init(bufa_->seed) */
	chaser_t ch; memset(&ch, 0, sizeof(ch));
	ch.probes = counts ? &counts->rstarts : NULL;
	ch.offTidx = OFF_MASK;
	v1_orient_t O[2];
	memset(O, 0, sizeof(O));
	O[0].drL = fw1 ? d1Fw : d1Rc; O[0].drR = fw2 ? d2Fw : d2Rc;       /* aligner.h:670-682 */
	O[1].drL = fw2 ? d2Rc : d2Fw; O[1].drR = fw1 ? d1Rc : d1Fw;       /* aligner.h:684-696 */
	pairset_t pairs_fw = {0, 0, 0}, pairs_rc = {0, 0, 0};
	range_t* found = (range_t*)malloc(sizeof(range_t));
	const uint32_t qlen1 = rd[0].len, qlen2 = rd[1].len;
	const uint32_t symCeil = pol->mhits;            /* "mhits, // for symCeiling" (ebwt_search.cpp:1275) */
	uint32_t mixedAttempts = 0;
	int done = 0, doneFw = 0, doneFwFirst = 1, o = 0;
	const int dbg = getenv("BTO_V1_DEBUG") != NULL;
#define V1LOG(...) do { if (dbg) fprintf(stderr, __VA_ARGS__); } while (0)
	while (!done) {
		/* ---- advance() (aligner.h:815-847) ---- */
		if (doneFw && doneFwFirst) { o = 1; doneFwFirst = 0; mixedAttempts = 0; }
		v1_orient_t* X = &O[o];
		if ((X->chaseL || X->chaseR) && ch.offTidx == OFF_MASK && !ch.done) { chaser_advance(&ch, &env); continue; }
		/* ---- advanceOrientation(pairFw = !doneFw) (aligner.h:1091-1320) ---- */
		const int pairFw = !doneFw;
		int* donePair = (o == 0) ? &doneFw : &done;
		int returned = 0;
		for (int side = 0; side < 2 && !returned; side++) {
			/* side 0: a row of L's range is being chased; side 1: of R's */
			int* chaseMe = side == 0 ? &X->chaseL : &X->chaseR;
			if (!*chaseMe) continue;
			if (side == 1 && X->chaseL) break;      /* "else if" */
			driver_t* drMe = side == 0 ? X->drL : X->drR;
			driver_t* drOther = side == 0 ? X->drR : X->drL;
			int* delayedOther = side == 0 ? &X->delayedR : &X->delayedL;
			int* chaseOther = side == 0 ? &X->chaseR : &X->chaseL;
			if (ch.offTidx != OFF_MASK) {
				if (!done) {
					/* mixed mode always (overThresh || dontReconcile_): resolveOutstandingInRef (aligner.h:951-1086) */
					const range_t* range = drv_range(drMe);
					const int off1 = side == 0 ? pairFw : !pairFw;
					const uint32_t tidx = ch.offTidx, toff = ch.offToff;
					int ret = 0;
					const int matchRight = off1 ? !doneFw : doneFw;
					int fw = off1 ? fw2 : fw1;
					if (doneFw) fw = !fw;
					const read_t* om = &rd[off1 ? 1 : 0];                      /* the outstanding mate */
					const uint8_t* oseq = fw ? om->pat[1][1] : om->pat[0][1];   /* patFw : patRc */
					const uint8_t* oqual = fw ? om->qual[0] : om->qual[1];
					const uint32_t qlen = om->len, alen = off1 ? rd[0].len : rd[1].len;
					const int minins = pol->min_ins, maxins = pol->max_ins;
					if ((uint32_t)maxins > (qlen > alen ? qlen : alen)) {
						uint32_t begin, end;
						const uint32_t insDiff = (uint32_t)(maxins - minins);
						const int contain = pol->allow_contain;
						if (matchRight) {
							end = toff + (uint32_t)maxins;
							begin = toff + (contain ? 0 : 1);
							if (!contain && qlen < alen) begin += alen - qlen;
							if (end > insDiff + qlen) { uint32_t b2 = end - insDiff - qlen; if (b2 > begin) begin = b2; }
							if (refs->approxLen[tidx] < end) end = refs->approxLen[tidx];
							if (refs->approxLen[tidx] < begin) begin = refs->approxLen[tidx];
						} else {
							begin = (toff + alen < (uint32_t)maxins) ? 0 : toff + alen - (uint32_t)maxins;
							const uint32_t mi = alen < qlen ? alen : qlen;
							if (contain) end = toff + alen;                     /* V1: no "- 1" (aligner.h:1046) */
							else {
								end = toff + mi - 1;
								const uint32_t e2 = toff + alen - (uint32_t)minins + qlen - 1;
								if (e2 < end) end = e2;
								if (toff + alen + qlen < (uint32_t)minins + 1) end = 0;
							}
						}
						/* "if(end - begin < qlen) return false" is unsigned there; end < begin would send the
						 * reference's find() a wrapped spread (it asserts end > begin): not restated */
						if (end >= begin && end - begin >= qlen) {
							uint32_t result = 0;
							if (ref_find_one(pol, refs->seq[tidx], oseq, oqual, qlen, tidx, begin, end, fw,
							                 doneFw ? &pairs_rc : &pairs_fw, toff, found, &result)) {
								range_t* r = found;
								r->fw = fw; r->cost = (uint16_t)(r->cost | (r->stratum << 14));
								r->mate1 = !off1; r->top = range->top; r->bot = range->bot;
								const int ebwtLFw = matchRight ? (range->ebwt->fw != 0) : 1;
								const int ebwtRFw = matchRight ? 1 : (range->ebwt->fw != 0);
								const range_t* rL = matchRight ? range : r;
								const range_t* rR = matchRight ? r : range;
								const uint32_t up = matchRight ? toff : result, dn = matchRight ? result : toff;
								/* report (aligner.h:856-936): upstream mate first; pairFw = !doneFw_ */
								const int pf = !doneFw;
								const uint32_t oms = (rL->bot - rL->top < rR->bot - rR->top ? rL->bot - rL->top : rR->bot - rR->top) - 1;
								const uint32_t lenL = pf ? rd[0].len : rd[1].len, lenR = pf ? rd[1].len : rd[0].len;
								ret = al_report_ex(&sink, rL, tidx, up, lenL, ebwtLFw, pf ? 1 : 2, oms);
								if (!ret) ret = al_report_ex(&sink, rR, tidx, dn, lenR, ebwtRFw, pf ? 2 : 1, oms);
							}
						}
					}
					V1LOG("[v1] o=%d side=%d attempt at %u:%u -> ret=%d stored=%d\n", o, side, tidx, toff, ret, sink.stored);
					done = ret;
					if (++mixedAttempts > (uint32_t)pol->pair_tries) { *donePair = 1; returned = 1; break; }
				}
				ch.offTidx = OFF_MASK;                  /* rchase_->reset() */
			} else {
				/* the chaser has been through the whole range */
				*chaseMe = 0;
				if (drMe) drMe->foundRange = 0;
				if (*delayedOther) {
					const range_t* r = drv_range(drOther);
					const uint32_t qlen = side == 0 ? (doneFw ? qlen1 : qlen2) : (doneFw ? qlen2 : qlen1);
					chaser_set_top_bot(&ch, r->top, r->bot, qlen, &alRnd, r->ebwt, &env);
					*chaseOther = 1; *delayedOther = 0;
				}
			}
			break;
		}
		if (returned) continue;
		if (!done && !*donePair && !X->chaseL && !X->chaseR) {
			/* look for more ranges for whichever mate has fewer candidates */
			int side;
			if ((X->offsLsz < X->offsRsz || V1_DONE(X->drR)) && !V1_DONE(X->drL)) side = 0;
			else if (!V1_DONE(X->drR)) side = 1;
			else { V1LOG("[v1] o=%d both drivers done btCnt=%d\n", o, btCnt ? *btCnt : -1); *donePair = 1; continue; }
			driver_t* drMe = side == 0 ? X->drL : X->drR;
			driver_t* drOther = side == 0 ? X->drR : X->drL;
			uint32_t* szMe = side == 0 ? &X->offsLsz : &X->offsRsz;
			uint32_t* szOther = side == 0 ? &X->offsRsz : &X->offsLsz;
			int* delayedMe = side == 0 ? &X->delayedL : &X->delayedR;
			int* delayedOther = side == 0 ? &X->delayedR : &X->delayedL;
			int* chaseMe = side == 0 ? &X->chaseL : &X->chaseR;
			int* chaseOther = side == 0 ? &X->chaseR : &X->chaseL;
			if (V1_DONE(drOther) && *szOther == 0) { V1LOG("[v1] o=%d give up (other done, 0 cand) side=%d btCnt=%d\n", o, side, btCnt ? *btCnt : -1); *donePair = 1; continue; }     /* no pair in this orientation */
			if (!drMe->foundRange) drv_advance(drMe, ADV_FOUND_RANGE);
			if (drMe->foundRange) {
				const range_t* rm = drv_range(drMe);
				V1LOG("[v1] o=%d side=%d found range [%u,%u) cost=%u btCnt=%d\n", o, side, rm->top, rm->bot, rm->cost, btCnt ? *btCnt : -1);
				*szMe += rm->bot - rm->top;
				if (*szOther == 0 && *szMe > 3) {       /* (!dontReconcile_ || sz > 3), dontReconcile_ == true */
					*delayedMe = 1;
				} else {
					if (*szMe > symCeil && *szOther > symCeil) { *donePair = 1; continue; }
					if (*delayedOther && *szOther < *szMe) {
						/* first range for both mates: chase the smaller one first */
						*delayedOther = 0; *delayedMe = 1; *chaseOther = 1;
						const range_t* r = drv_range(drOther);
						const uint32_t qlen = side == 0 ? (doneFw ? qlen1 : qlen2) : (doneFw ? qlen2 : qlen1);
						chaser_set_top_bot(&ch, r->top, r->bot, qlen, &alRnd, r->ebwt, &env);
					} else {
						*chaseMe = 1;
						const uint32_t qlen = side == 0 ? (doneFw ? qlen2 : qlen1) : (doneFw ? qlen1 : qlen2);
						chaser_set_top_bot(&ch, rm->top, rm->bot, qlen, &alRnd, rm->ebwt, &env);
					}
				}
			}
		}
	}
	if (budget < 0) st |= BT_ST_OVERFLOW;
	for (int i = 0; i < 4; i++) if (all[i]) drv_free(all[i]);
	free(rd); free(found); free(pairs_fw.a); free(pairs_rc.a);
	if (sink.dropped && sink.hitsForThisRead <= sink.max) st |= BT_ST_HITCAP;
	if (n_hits_total) *n_hits_total = sink.hitsForThisRead;
	if (status) *status = st;
	if (sink.strata) for (int i = 0; i < sink.stored; i++) sink.hits[i].oms = (uint32_t)sink.stored / 2u - 1u;
	if (sink.hitsForThisRead > sink.max) return pol->sample_max ? sink.stored : 0;
	int n = sink.stored;
	if ((uint32_t)n > sink.n) n = (int)sink.n;
	return n;
}
