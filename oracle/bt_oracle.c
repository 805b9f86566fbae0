/*
 * bt_oracle.c -- TEST INFRASTRUCTURE ONLY (see bt_oracle.h).
 *
 * A deliberately simple, recursive, one-read-at-a-time CPU restatement of the reference's
 * FM-index search path, written from the behaviour of BenLangmead/bowtie v1.3.1.  Every
 * function cites the reference code whose behaviour it restates.  It shares no code with the
 * HIP kernels in bowtie_amd/csrc (those are an iterative, batched state machine).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bt_oracle.h"

#define OFF_MASK 0xffffffffu

/* ------------------------------------------------------------------------------------------
 * Index loading: Ebwt::readIntoMemory, ebwt.h:2926-3272 (.1.ebwt) and 3274-3421 (.2.ebwt);
 * geometry: EbwtParams::init, ebwt.h:138-184; refnames: readEbwtRefnames, ebwt.h:3452-3531;
 * zOff -> (byte,bp): Ebwt::postReadInit, ebwt.h:1043-1059.
 * ---------------------------------------------------------------------------------------- */
static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int bto_index_load(const char* base, int fw, bto_index* ix)
{
	char path[4096];
	memset(ix, 0, sizeof(*ix));
	ix->fw = fw;
	snprintf(path, sizeof(path), "%s.1.ebwt", base);
	FILE* f1 = fopen(path, "rb");
	if (!f1) return BT_ERR_IO;
	uint32_t one; int32_t lineRate, linesPerSide, offRate, ftabChars, flags;
	if (rd(f1, &one, 4) || one != 1) { fclose(f1); return BT_ERR_FORMAT; }
	if (rd(f1, &ix->len, 4) || rd(f1, &lineRate, 4) || rd(f1, &linesPerSide, 4) ||
	    rd(f1, &offRate, 4) || rd(f1, &ftabChars, 4) || rd(f1, &flags, 4)) { fclose(f1); return BT_ERR_IO; }
	if (lineRate != 6 || linesPerSide != 1) { fclose(f1); return BT_ERR_FORMAT; }
	ix->bwtLen = ix->len + 1;
	uint32_t bwtSz = ix->len / 4 + 1;
	ix->sideSz = 64; ix->sideBwtSz = 56; ix->sideBwtLen = 224;
	uint32_t numSidePairs = (bwtSz + 2 * 56 - 1) / (2 * 56);
	ix->numSides = numSidePairs * 2;
	ix->ebwtTotLen = numSidePairs * 128;
	ix->ftabChars = (uint32_t)ftabChars;
	ix->ftabLen = (1u << (2 * ftabChars)) + 1;
	ix->eftabLen = 2 * (uint32_t)ftabChars;
	ix->offRate = (uint32_t)offRate;
	ix->offMask = OFF_MASK << offRate;
	ix->offsLen = (ix->bwtLen + (1u << offRate) - 1) >> offRate;
	if (rd(f1, &ix->nPat, 4)) { fclose(f1); return BT_ERR_IO; }
	ix->plen = (uint32_t*)malloc(4 * (size_t)ix->nPat);
	if (rd(f1, ix->plen, 4 * (size_t)ix->nPat) || rd(f1, &ix->nFrag, 4)) { fclose(f1); return BT_ERR_IO; }
	ix->rstarts = (uint32_t*)malloc(12 * (size_t)ix->nFrag);
	ix->ebwt = (uint8_t*)malloc(ix->ebwtTotLen);
	ix->ftab = (uint32_t*)malloc(4 * (size_t)ix->ftabLen);
	ix->eftab = (uint32_t*)malloc(4 * (size_t)ix->eftabLen);
	if (rd(f1, ix->rstarts, 12 * (size_t)ix->nFrag) || rd(f1, ix->ebwt, ix->ebwtTotLen) ||
	    rd(f1, &ix->zOff, 4) || rd(f1, ix->fchr, 20) || rd(f1, ix->ftab, 4 * (size_t)ix->ftabLen) ||
	    rd(f1, ix->eftab, 4 * (size_t)ix->eftabLen)) { fclose(f1); return BT_ERR_IO; }
	/* reference names: '\n'-separated, '\0'-terminated */
	ix->refnames = (char**)calloc(ix->nPat + 1, sizeof(char*));
	{
		size_t cap = 256, n = 0; char* buf = (char*)malloc(cap); uint32_t k = 0; int c;
		while ((c = fgetc(f1)) != EOF && c != 0) {
			if (c == '\n') {
				buf[n] = 0;
				if (k < ix->nPat) ix->refnames[k++] = strdup(buf);
				n = 0;
			} else {
				if (n + 2 > cap) { cap *= 2; buf = (char*)realloc(buf, cap); }
				buf[n++] = (char)c;
			}
		}
		if (n > 0 && k < ix->nPat) { buf[n] = 0; ix->refnames[k++] = strdup(buf); }
		free(buf);
	}
	fclose(f1);
	snprintf(path, sizeof(path), "%s.2.ebwt", base);
	FILE* f2 = fopen(path, "rb");
	if (!f2) return BT_ERR_IO;
	ix->offs = (uint32_t*)malloc(4 * (size_t)ix->offsLen);
	if (rd(f2, &one, 4) || one != 1 || rd(f2, ix->offs, 4 * (size_t)ix->offsLen)) { fclose(f2); return BT_ERR_IO; }
	fclose(f2);
	/* postReadInit */
	uint32_t sideNum = ix->zOff / 224, sideCharOff = ix->zOff % 224;
	uint32_t byteOff = sideCharOff >> 2; int bp = (int)(sideCharOff & 3);
	if ((sideNum & 1) == 0) { byteOff = 56 - byteOff - 1; bp = 3 - bp; }
	ix->zEbwtByteOff = byteOff + sideNum * 64;
	ix->zEbwtBpOff = bp;
	return BT_OK;
}

void bto_index_free(bto_index* ix)
{
	free(ix->plen); free(ix->rstarts); free(ix->ebwt); free(ix->ftab); free(ix->eftab); free(ix->offs);
	if (ix->refnames) { for (uint32_t i = 0; i < ix->nPat; i++) free(ix->refnames[i]); free(ix->refnames); }
	memset(ix, 0, sizeof(*ix));
}

/* ------------------------------------------------------------------------------------------
 * Rank / LF.  SideLocus::initFromRow ebwt.h:1469-1497; countUpToEx ebwt.h:1963-2027;
 * countFwSideEx ebwt.h:2081-2129; countBwSideEx ebwt.h:2184-2226; rowL ebwt.h:1696.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t sideByteOff; int fw, by, bp; } locus_t;

static void locus_init(locus_t* l, uint32_t row)
{
	uint32_t sideNum = row / 224, charOff = row % 224;
	l->sideByteOff = sideNum * 64;
	l->fw = (int)(sideNum & 1);
	l->by = (int)(charOff >> 2);
	l->bp = (int)(charOff & 3);
	if (!l->fw) { l->by = 56 - l->by - 1; l->bp ^= 3; }
}

static int locus_rowL(const bto_index* ix, const locus_t* l)
{
	return (ix->ebwt[l->sideByteOff + (uint32_t)l->by] >> (2 * l->bp)) & 3;
}

static void count_up_to(const uint8_t* side, int by, int bp, uint32_t a[4])
{
	for (int i = 0; i < by; i++)
		for (int k = 0; k < 4; k++) a[(side[i] >> (2 * k)) & 3]++;
	for (int k = 0; k < bp; k++) a[(side[by] >> (2 * k)) & 3]++;
}

static void rank4_locus(const bto_index* ix, const locus_t* l, uint32_t lf[4])
{
	const uint8_t* side = ix->ebwt + l->sideByteOff;
	uint32_t a[4] = {0, 0, 0, 0};
	count_up_to(side, l->by, l->bp, a);
	uint32_t pos = l->sideByteOff + (uint32_t)l->by;
	const uint32_t *ac, *gt;
	if (l->fw) {
		/* '$' is stored as an A but must not count as one (ebwt.h:2099-2107) */
		if (l->sideByteOff <= ix->zEbwtByteOff && pos >= ix->zEbwtByteOff &&
		    (pos > ix->zEbwtByteOff || l->bp > ix->zEbwtBpOff)) a[0]--;
		ac = (const uint32_t*)(side - 8);
		gt = (const uint32_t*)(side + 64 - 8);
		lf[0] = a[0] + ac[0] + ix->fchr[0];
		lf[1] = a[1] + ac[1] + ix->fchr[1];
		lf[2] = a[2] + gt[0] + ix->fchr[2];
		lf[3] = a[3] + gt[1] + ix->fchr[3];
	} else {
		a[locus_rowL(ix, l)]++;
		if (l->sideByteOff <= ix->zEbwtByteOff && pos >= ix->zEbwtByteOff &&
		    (pos > ix->zEbwtByteOff || l->bp >= ix->zEbwtBpOff)) a[0]--;
		ac = (const uint32_t*)(side + 64 - 8);
		gt = (const uint32_t*)(side + 128 - 8);
		lf[0] = ac[0] - a[0] + ix->fchr[0];
		lf[1] = ac[1] - a[1] + ix->fchr[1];
		lf[2] = gt[0] - a[2] + ix->fchr[2];
		lf[3] = gt[1] - a[3] + ix->fchr[3];
	}
}

void bto_rank4(const bto_index* ix, uint32_t row, uint32_t lf[4])
{
	locus_t l; locus_init(&l, row); rank4_locus(ix, &l, lf);
}

int bto_rowL(const bto_index* ix, uint32_t row)
{
	locus_t l; locus_init(&l, row); return locus_rowL(ix, &l);
}

/* ftabHi / ftabLo, ebwt.h:985-1034 */
uint32_t bto_ftab_hi(const bto_index* ix, uint32_t i)
{
	if (ix->ftab[i] <= ix->len) return ix->ftab[i];
	return ix->eftab[(ix->ftab[i] ^ OFF_MASK) * 2 + 1];
}
uint32_t bto_ftab_lo(const bto_index* ix, uint32_t i)
{
	if (ix->ftab[i] <= ix->len) return ix->ftab[i];
	return ix->eftab[(ix->ftab[i] ^ OFF_MASK) * 2];
}

/* reportChaseOne's walk, ebwt.h:2727-2746 */
uint32_t bto_chase(const bto_index* ix, uint32_t row, uint32_t* jumps_out)
{
	uint32_t jumps = 0, i = row;
	while ((i & ix->offMask) != i && i != ix->zOff) {
		uint32_t lf[4];
		bto_rank4(ix, i, lf);
		i = lf[bto_rowL(ix, i)];           /* mapLF(l), ebwt.h:2420 */
		jumps++;
	}
	if (jumps_out) *jumps_out = jumps;
	return (i == ix->zOff) ? jumps : ix->offs[i >> ix->offRate] + jumps;
}

/* Ebwt::restore (ebwt.h:2793-2824): invert the BWT into the joined text (codes 0..3).  For the
 * mirror index this yields the reversed joined text. */
void bto_restore_text(const bto_index* ix, uint8_t* out)
{
	uint32_t i = ix->len, jumps = 0;     /* the row of the suffix "$" (sorts last) */
	while (i != ix->zOff) {
		uint32_t lf[4];
		int c = bto_rowL(ix, i);
		bto_rank4(ix, i, lf);
		out[ix->len - 1 - jumps] = (uint8_t)c;
		i = lf[c];
		jumps++;
	}
}

/* joinedToTextOff, ebwt.h:2569-2629 */
int bto_joined_to_text_cnt(const bto_index* ix, uint32_t qlen, uint32_t off,
                           uint32_t* tidx, uint32_t* toff, uint32_t* tlen, uint64_t* probes)
{
	uint32_t top = 0, bot = ix->nFrag;
	for (;;) {
		uint32_t elt = top + ((bot - top) >> 1);
		if (probes) (*probes)++;
		uint32_t lower = ix->rstarts[elt * 3];
		uint32_t upper = (elt == ix->nFrag - 1) ? ix->len : ix->rstarts[(elt + 1) * 3];
		uint32_t fraglen = upper - lower;
		if (lower <= off) {
			if (upper > off) {
				if (off + qlen > upper) { *tidx = OFF_MASK; return 0; }
				*tidx = ix->rstarts[elt * 3 + 1];
				uint32_t fragoff = off - ix->rstarts[elt * 3];
				if (!ix->fw) { fragoff = fraglen - fragoff - 1; fragoff -= (qlen - 1); }
				*toff = fragoff + ix->rstarts[elt * 3 + 2];
				break;
			}
			top = elt;
		} else {
			bot = elt;
		}
	}
	if (tlen) *tlen = ix->plen[*tidx];
	return 1;
}

int bto_joined_to_text(const bto_index* ix, uint32_t qlen, uint32_t off,
                       uint32_t* tidx, uint32_t* toff, uint32_t* tlen)
{
	return bto_joined_to_text_cnt(ix, qlen, off, tidx, toff, tlen, NULL);
}

/* genRandSeed, pat.cpp:21-57 */
uint32_t bto_rand_seed(const uint8_t* seq, const uint8_t* qual, int len,
                       const char* name, int namelen, uint32_t global_seed)
{
	uint32_t rseed = (global_seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	for (int i = 0; i < len; i++) rseed ^= ((uint32_t)seq[i] << ((i & 15) << 1));
	for (int i = 0; i < len; i++) rseed ^= ((uint32_t)qual[i] << ((i & 3) << 3));
	for (int i = 0; i < namelen; i++) rseed ^= ((uint32_t)(uint8_t)name[i] << ((i & 3) << 3));
	return rseed;
}

/* ------------------------------------------------------------------------------------------
 * Per-read sink: NGoodHitSinkPerThread::reportHit hit.h:969-985, AllHitSinkPerThread::reportHit
 * hit.h:1201-1209, finishRead hit.h:741-786.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t n, max;              /* _n, _max */
	uint32_t hitsForThisRead;
	bto_hit* hits; int cap; int stored;
	int all, dropped;
} sink_t;

static int sink_report(sink_t* s, const bto_hit* h)
{
	s->hitsForThisRead++;
	if (s->hitsForThisRead > s->max) return 1;
	if (s->stored < s->cap) s->hits[s->stored++] = *h;      /* bufferHit */
	else if ((uint32_t)s->stored < s->n) s->dropped = 1;
	if (s->all) return 0;
	if (s->hitsForThisRead == s->n && (s->max == 0xffffffffu || s->max < s->n)) return 1;
	return 0;
}

/* mmPenalty / phredCharToPhredQual / qualRounds, qual.h:15,61-67, qual.cpp:4-32 */
static uint8_t phred_of(uint8_t c) { return c >= 33 ? (uint8_t)(c - 33) : 0; }
static uint8_t mm_penalty(int maq, uint8_t q)
{
	if (!maq) return q;
	if (q < 5) return 0;
	if (q < 15) return 10;
	if (q < 25) return 20;
	return 30;
}

/* ------------------------------------------------------------------------------------------
 * GreedyDFSRangeSource, ebwt_search_backtrack.h:23-1779
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint16_t pos; uint8_t oldBase, newBase; } mut_t;

typedef struct {
	const bto_index* ix;
	/* query as selected by setQuery (ebwt_search_backtrack.h:90-140) */
	uint8_t  qry[BTO_MAXLEN], qual[BTO_MAXLEN];
	uint32_t fullLen, qlen;
	int      readFw;               /* EbwtSearchParams::_fw */
	/* setOffs (:162-176) */
	uint32_t d5, d3, unrev, r1, r2, r3;
	/* ctor params (:28-79) */
	uint32_t qualThresh, maxBts, reportPartials;
	int      reportExacts, considerQuals, halfAndHalf, maqPenalty;
	/* state */
	uint32_t rnd;                  /* RandomSource::last (random_source.h:45-54) */
	uint32_t numBts; int bailed;
	uint32_t *pairs; uint8_t *elims;
	uint32_t mms[BTO_MAXLEN]; uint8_t refcs[BTO_MAXLEN];
	int      nmuts; mut_t muts[3];
	uint64_t partials[4096]; int npartials;
	sink_t*  sink;
	bt_op_counts* cnt;
} dfs_t;

static uint32_t rnd_u32(dfs_t* s)
{
	uint32_t ret;
	s->rnd = 1664525u * s->rnd + 1013904223u;
	ret = s->rnd >> 16;
	s->rnd = 1664525u * s->rnd + 1013904223u;
	ret ^= s->rnd;
	return ret;
}

/* setQuery: pick patFw|patRc|patFwRev|patRcRev and qual|qualRev by (readFw, index fw), and
 * re-seed the RNG (:90-140).  Read::constructRevComps/constructReverses, read.h:119-133. */
static void dfs_set_query(dfs_t* s, const bto_index* ix, int readFw,
                          const uint8_t* seq, const uint8_t* qual, uint32_t len, uint32_t seed)
{
	s->ix = ix; s->readFw = readFw; s->fullLen = s->qlen = len;
	int rev = (ix->fw != 0) != (readFw != 0);
	for (uint32_t i = 0; i < len; i++) {
		uint32_t j = rev ? len - 1 - i : i;
		uint8_t c = seq[j];
		if (!readFw && c < 4) c ^= 3;
		s->qry[i] = c;
		s->qual[i] = qual[j];
	}
	s->rnd = seed;
	s->nmuts = 0;
}

static void dfs_set_offs(dfs_t* s, uint32_t d5, uint32_t d3, uint32_t unrev, uint32_t r1, uint32_t r2, uint32_t r3)
{
	s->d5 = d5; s->d3 = d3; s->unrev = unrev; s->r1 = r1; s->r2 = r2; s->r3 = r3;
}

#define PTOP(p, d, c) ((p)[(d) * 8 + (c)])
#define PBOT(p, d, c) ((p)[(d) * 8 + (c) + 4])

/* Ebwt::report + EbwtSearchParams::reportHit (ebwt.h:2635-2682, 1288-1405): returns 1 iff the
 * sink says stop. */
static int dfs_report_row(dfs_t* s, uint32_t numMms, uint32_t row, uint32_t top, uint32_t bot,
                          int stratum, uint16_t cost)
{
	const bto_index* ix = s->ix;
	uint32_t jumps = 0;
	uint32_t off = bto_chase(ix, row, &jumps);
	if (s->cnt) { s->cnt->chase += jumps; s->cnt->offs++; }
	uint32_t tidx, toff, tlen;
	if (!bto_joined_to_text_cnt(ix, s->qlen, off, &tidx, &toff, &tlen, s->cnt ? &s->cnt->rstarts : NULL)) return 0;
	bto_hit h;
	memset(&h, 0, sizeof(h));
	h.tidx = tidx; h.toff = toff; h.oms = bot - top - 1;
	h.cost = cost; h.stratum = (uint8_t)stratum; h.fw = (uint8_t)s->readFw;
	if (numMms > BTO_MAXMM) numMms = BTO_MAXMM;   /* tests never get here */
	h.nmm = (uint16_t)numMms;
	int flip = (ix->fw != 0) != (s->readFw != 0);
	for (uint32_t i = 0; i < numMms; i++) {
		uint32_t pos = flip ? s->qlen - s->mms[i] - 1 : s->mms[i];
		h.mm[i] = (uint16_t)(pos | ((uint32_t)(s->refcs[i] & 3) << 12));
	}
	/* Hit::mms is a bitset: order by position */
	for (uint32_t i = 1; i < numMms; i++) {
		uint16_t v = h.mm[i]; int j = (int)i - 1;
		while (j >= 0 && BT_MM_POS(h.mm[j]) > BT_MM_POS(v)) { h.mm[j + 1] = h.mm[j]; j--; }
		h.mm[j + 1] = v;
	}
	return sink_report(s->sink, &h);
}

/* reportFullAlignment (:1522-1565) */
static int dfs_report_full(dfs_t* s, uint32_t stackDepth, uint32_t top, uint32_t bot, int stratum, uint16_t cost)
{
	if (stackDepth == 0 && !s->reportExacts) return 0;
	uint32_t spread = bot - top;
	uint32_t r = rnd_u32(s);
	if (s->ix->wide) { uint64_t r64 = ((uint64_t)r << 32) | rnd_u32(s); r = (uint32_t)(r64 % spread); }   /* nextU<TIndexOffU>() */
	else r %= spread;
	r += top;
	for (uint32_t i = 0; i < spread; i++) {
		uint32_t ri = r + i;
		if (ri >= bot) ri -= spread;
		if (dfs_report_row(s, stackDepth, ri, top, bot, stratum, cost)) return 1;
	}
	return 0;
}

/* reportPartial (:1571-1655); PartialAlignment bit layout ebwt_search_util.h:37-88 */
static void dfs_report_partial(dfs_t* s, uint32_t stackDepth)
{
	uint64_t al = 0xffffffffffffffffull;
	uint64_t pos[3] = {0xffff, 0xffff, 0xffff}, chr[3] = {3, 3, 3};
	for (uint32_t k = 0; k < stackDepth && k < 3; k++) { pos[k] = s->mms[k] & 0xffff; chr[k] = s->refcs[k] & 3; }
	/* unused char slots keep the all-ones initialisation */
	al = (pos[0]) | (pos[1] << 16) | (pos[2] << 32) | (chr[0] << 48) | (chr[1] << 50) | (chr[2] << 52)
	   | (0xffull << 54) | (3ull << 62);
	if (s->npartials < 4096) s->partials[s->npartials++] = al;
}

/* calcStratum (:1164-1177) */
static int dfs_stratum(const dfs_t* s, uint32_t stackDepth)
{
	int stratum = 0;
	for (uint32_t i = 0; i < stackDepth; i++)
		if (s->mms[i] >= (s->qlen - s->r3)) stratum++;
	return stratum;
}

/* reportAlignment (:1455-1513) incl. undo/promote/re-apply of partial-alignment mutations
 * (:1368-1446) */
static int dfs_report_alignment(dfs_t* s, uint32_t stackDepth, uint32_t top, uint32_t bot, uint16_t cost)
{
	if (s->reportPartials) {
		if (stackDepth > 0) dfs_report_partial(s, stackDepth);
		return 0;
	}
	int stratum = stackDepth > 0 ? dfs_stratum(s, stackDepth) : 0;
	if (s->nmuts > 0) {
		for (int i = 0; i < s->nmuts; i++) {
			s->mms[stackDepth + (uint32_t)i] = s->muts[i].pos;
			s->refcs[stackDepth + (uint32_t)i] = s->muts[i].newBase;
		}
		stratum += s->nmuts;
		cost |= (uint16_t)(stratum << 14);
		return dfs_report_full(s, stackDepth + (uint32_t)s->nmuts, top, bot, stratum, cost);
	}
	cost |= (uint16_t)(stratum << 14);
	return dfs_report_full(s, stackDepth, top, bot, stratum, cost);
}

/* hhCheckTop (:1200-1275) */
static int dfs_hh_check_top(const dfs_t* s, uint32_t stackDepth, uint32_t d)
{
	if (d == s->d5) {
		if (stackDepth == 0) return 0;     /* both the 2- and 3-mismatch flavours */
	} else if (d == s->d3) {
		if (s->r3 == s->r2) {
			if (stackDepth < 2) return 0;
		} else {
			int lo = 0;
			for (uint32_t i = 0; i < stackDepth; i++) {
				uint32_t dd = s->qlen - s->mms[i] - 1;
				if (dd >= s->d5 && dd < s->d3) lo++;
			}
			if (lo == 0) return 0;
		}
	}
	return 1;
}

static void dfs_lf_pair(dfs_t* s, uint32_t top, uint32_t bot, uint32_t tops[4], uint32_t bots[4])
{
	if (s->cnt && top / 448 == bot / 448) s->cnt->same_pair++;
	bto_rank4(s->ix, top, tops);
	bto_rank4(s->ix, bot, bots);
}

/* the recursive backtrack (:363-1091) */
static int dfs_frame(dfs_t* s, uint32_t stackDepth, uint32_t depth,
                     uint32_t unrevOff, uint32_t oneRevOff, uint32_t twoRevOff, uint32_t threeRevOff,
                     uint32_t top, uint32_t bot, uint32_t ham, uint32_t iham,
                     uint32_t* pairs, uint8_t* elims, int disableFtab)
{
	const bto_index* ix = s->ix;
	const uint32_t qlen = s->qlen;
	if (s->cnt) s->cnt->frames++;
	if (s->halfAndHalf) {
		if (s->maxBts > 0 && s->numBts == s->maxBts) { s->bailed = 1; return 0; }
		s->numBts++;
	}
	uint32_t altNum = 0, eligibleNum = 0, eligibleSz = 0;
	uint32_t eli = 0, eltop = 0, elbot = 0, elham = ham; int elcint = 0, elignore = 1;
	uint32_t lowAltQual = 0xff;
	uint32_t d = depth;
	uint32_t cur = qlen - d - 1;
	while (cur < qlen) {
		if (s->halfAndHalf && !dfs_hh_check_top(s, stackDepth, d)) return 0;
		int curIsEligible = 0, curOverridesEligible = 0;
		int c = s->qry[cur];
		uint8_t q = phred_of(s->qual[cur]);
		int curIsAlternative = (d >= unrevOff) &&
			(!s->considerQuals || (ham + mm_penalty(s->maqPenalty, q) <= s->qualThresh));
		if (curIsAlternative) {
			if (s->considerQuals) {
				if (q < lowAltQual) { curIsEligible = 1; curOverridesEligible = 1; }
				else if (q == lowAltQual) curIsEligible = 1;
			} else curIsEligible = 1;
		}
		/* rtop/rbot: the real range coming into this position (the reference keeps it in the
		 * SideLocus pair while it overwrites top/bot with 1 for an N) */
		uint32_t rtop = top, rbot = bot;
		if (c == 4 && d > 0) top = bot = 1;
		if (top == 0 && bot == 0) {
			PTOP(pairs, 0, 0) = ix->fchr[0];
			PBOT(pairs, 0, 0) = PTOP(pairs, 0, 1) = ix->fchr[1];
			PBOT(pairs, 0, 1) = PTOP(pairs, 0, 2) = ix->fchr[2];
			PBOT(pairs, 0, 2) = PTOP(pairs, 0, 3) = ix->fchr[3];
			PBOT(pairs, 0, 3) = ix->fchr[4];
			if (c < 4) { top = PTOP(pairs, d, c); bot = PBOT(pairs, d, c); }
		} else if (curIsAlternative) {
			dfs_lf_pair(s, rtop, rbot, &pairs[d * 8], &pairs[d * 8 + 4]);
			if (s->cnt) s->cnt->lfex++;
			if (c < 4) { top = PTOP(pairs, d, c); bot = PBOT(pairs, d, c); }
		} else if (c < 4) {
			uint32_t tl[4], bl[4];
			if (top + 1 == bot) {
				/* mapLF1 (ebwt.h:2494-2512) */
				if (s->cnt) s->cnt->lf1++;
				if (bto_rowL(ix, top) != c || top == ix->zOff) top = bot = OFF_MASK;
				else { bto_rank4(ix, top, tl); top = bot = tl[c]; bot++; }
			} else {
				if (s->cnt) s->cnt->lf2++;
				dfs_lf_pair(s, top, bot, tl, bl);
				top = tl[c]; bot = bl[c];
			}
		}
		elims[d] = (uint8_t)((c < 4) ? (1 << c) : 0);
		if (curIsAlternative) {
			for (int i = 0; i < 4; i++) {
				if (i == c) continue;
				uint32_t spread = PBOT(pairs, d, i) - PTOP(pairs, d, i);
				if (spread == 0) elims[d] |= (uint8_t)(1 << i);
				if (spread > 0 && ((elims[d] & (1 << i)) == 0)) {
					if (curIsEligible) {
						if (curOverridesEligible) {
							lowAltQual = q; eligibleNum = 0; eligibleSz = 0; curOverridesEligible = 0;
							eli = d; eltop = PTOP(pairs, d, i); elbot = PBOT(pairs, d, i);
							elham = mm_penalty(s->maqPenalty, q); elcint = i; elignore = 0;
						}
						eligibleSz += spread; eligibleNum++;
					}
					altNum++;
				}
			}
		}
		int backtrackDespiteMatch = 0, reportedPartial = 0;
		if (cur == 0 && top < bot && stackDepth < s->reportPartials && s->reportPartials > 0) {
			if (altNum > 0) backtrackDespiteMatch = 1;
			if (stackDepth > 0) { dfs_report_partial(s, stackDepth); reportedPartial = 1; }
		}
		int invalidExact = 0;
		if (cur == 0 && stackDepth == 0 && bot > top && !s->reportExacts) { invalidExact = 1; backtrackDespiteMatch = 1; }
		int mustBacktrack = 0, invalidHalfAndHalf = 0;
		if (s->halfAndHalf) {
			if (d == s->d5 - 1 && top < bot) {
				invalidHalfAndHalf = (stackDepth == 0);
				if (stackDepth == 0 && altNum > 0) { backtrackDespiteMatch = 1; mustBacktrack = 1; }
				else if (stackDepth == 0) return 0;
			} else if (d == s->d3 - 1 && top < bot) {
				uint32_t lo = 0, hi = 0;
				for (uint32_t i = 0; i < stackDepth; i++) {
					uint32_t dd = qlen - s->mms[i] - 1;
					if (dd < s->d5) hi++; else if (dd < s->d3) lo++;
				}
				invalidHalfAndHalf = (lo == 0 || hi == 0);
				if ((stackDepth < 2 || invalidHalfAndHalf) && altNum > 0) { mustBacktrack = 1; backtrackDespiteMatch = 1; }
				else if (stackDepth < 2) return 0;
			}
		}
		if (cur == 0 && bot > top && !invalidHalfAndHalf && !invalidExact && !reportedPartial) {
			if (dfs_report_alignment(s, stackDepth, top, bot, (uint16_t)ham)) return 1;
			top = bot;
		}
		while ((top == bot || backtrackDespiteMatch) && altNum > 0) {
			uint32_t i = d; int j = 0;
			uint32_t bttop = 0, btbot = 0, btham = ham; int btcint = 0;
			uint32_t icur;
			if (eligibleNum > 1 || elignore) {
				for (;; i--) {                       /* deepest (leftmost) eligible position first */
					icur = qlen - i - 1;
					uint8_t qi = phred_of(s->qual[icur]);
					if ((qi == lowAltQual || !s->considerQuals) && elims[i] != 15) {
						uint32_t posSz = 0;
						for (j = 0; j < 4; j++)
							if ((elims[i] & (1 << j)) == 0) posSz += PBOT(pairs, i, j) - PTOP(pairs, i, j);
						uint32_t r = rnd_u32(s) % posSz;
						for (j = 0; j < 4; j++) {
							if ((elims[i] & (1 << j)) == 0) {
								uint32_t spread = PBOT(pairs, i, j) - PTOP(pairs, i, j);
								if (r < spread) {
									bttop = PTOP(pairs, i, j); btbot = PBOT(pairs, i, j);
									btham += mm_penalty(s->maqPenalty, qi);
									btcint = j;
									break;
								}
								r -= spread;
							}
						}
						break;
					}
					if (i == depth) { fprintf(stderr, "oracle: no backtrack target found\n"); abort(); }
				}
			} else {
				i = eli; bttop = eltop; btbot = elbot; btham += elham; j = btcint = elcint;
			}
			icur = qlen - i - 1;
			uint32_t *newPairs = pairs + qlen * 8; uint8_t *newElims = elims + qlen;
			uint32_t btUnrev = unrevOff, btOne = oneRevOff, btTwo = twoRevOff, btThree = threeRevOff;
			if (i < oneRevOff)      { btUnrev = oneRevOff; btOne = twoRevOff; btTwo = threeRevOff; }
			else if (i < twoRevOff) { btOne = twoRevOff; btTwo = threeRevOff; }
			else if (i < threeRevOff) { btTwo = threeRevOff; }
			s->mms[stackDepth] = icur;
			s->refcs[stackDepth] = (uint8_t)btcint;
			int ret;
			if (i + 1 == qlen) {
				ret = dfs_report_alignment(s, stackDepth + 1, bttop, btbot, (uint16_t)btham);
			} else if (s->halfAndHalf && !disableFtab && s->r2 == s->r3 &&
			           i + 1 < ix->ftabChars && ix->ftabChars <= s->d5) {
				/* re-jump through the ftab with the substituted character (:908-952) */
				uint32_t ftabChars = ix->ftabChars;
				uint32_t ftabOff = s->qry[qlen - ftabChars];
				for (uint32_t jj = ftabChars - 1; jj > 0; jj--) {
					ftabOff <<= 2;
					if (qlen - jj == icur) ftabOff |= (uint32_t)btcint;
					else ftabOff |= s->qry[qlen - jj];
				}
				uint32_t ftabTop = bto_ftab_hi(ix, ftabOff), ftabBot = bto_ftab_lo(ix, ftabOff + 1);
				if (s->cnt) s->cnt->ftab++;
				if (ftabTop == ftabBot) ret = 0;
				else ret = dfs_frame(s, stackDepth + 1, ftabChars, btUnrev, btOne, btTwo, btThree,
				                     ftabTop, ftabBot, btham, iham, newPairs, newElims, 0);
			} else {
				ret = dfs_frame(s, stackDepth + 1, i + 1, btUnrev, btOne, btTwo, btThree,
				                bttop, btbot, btham, iham, newPairs, newElims, 0);
			}
			if (ret) return 1;
			if (s->bailed || (s->halfAndHalf && s->maxBts > 0 && s->numBts >= s->maxBts)) { s->bailed = 1; return 0; }
			elims[i] |= (uint8_t)(1 << j);
			eligibleSz -= (btbot - bttop);
			eligibleNum--;
			elignore = 1;
			altNum--;
			if (altNum == 0) return 0;
			if (eligibleNum == 0 && s->considerQuals) {
				/* re-scan the frame for the next-lowest-quality set of targets (:1004-1058) */
				lowAltQual = 0xff;
				for (uint32_t k = d; k >= depth && k <= qlen; k--) {
					uint32_t kcur = qlen - k - 1;
					uint8_t kq = phred_of(s->qual[kcur]);
					if (k < unrevOff) break;
					int kAlt = (ham + mm_penalty(s->maqPenalty, kq) <= s->qualThresh);
					int kOver = 0;
					if (kAlt) {
						if (kq < lowAltQual) kOver = 1;
						if (kq <= lowAltQual) {
							for (int l = 0; l < 4; l++) {
								if ((elims[k] & (1 << l)) == 0) {
									uint32_t spread = PBOT(pairs, k, l) - PTOP(pairs, k, l);
									if (kOver) {
										lowAltQual = kq; kOver = 0; eligibleNum = 0; eligibleSz = 0;
										eli = k; eltop = PTOP(pairs, k, l); elbot = PBOT(pairs, k, l);
										elham = mm_penalty(s->maqPenalty, kq); elcint = l; elignore = 0;
									}
									eligibleNum++; eligibleSz += spread;
								}
							}
						}
					}
					if (k == 0) break;   /* size_t wrap in the reference ends the loop via k <= qlen */
				}
			}
		}
		if (mustBacktrack || invalidHalfAndHalf || invalidExact) return 0;
		if (top == bot && altNum == 0) return 0;
		d++; cur--;
	}
	if (stackDepth >= s->reportPartials) return dfs_report_alignment(s, stackDepth, top, bot, (uint16_t)ham);
	return 0;
}

/* tallyNs (:1308-1341) */
static int dfs_tally_ns(const dfs_t* s, int* nsInFtab)
{
	int nsInSeed = 0;
	for (uint32_t i = 0; i < s->r3; i++) {
		if (s->qry[s->qlen - i - 1] == 4) {
			nsInSeed++;
			if (nsInSeed == 1) { if (i < s->unrev) return 0; }
			else if (nsInSeed == 2) { if (i < s->r1) return 0; }
			else if (nsInSeed == 3) { if (i < s->r2) return 0; }
			else return 0;
		}
	}
	*nsInFtab = 0;
	for (uint32_t i = 0; i < s->ix->ftabChars && i < s->qlen; i++)
		if (s->qry[s->qlen - i - 1] == 4) (*nsInFtab)++;
	return 1;
}

/* backtrack(ham) entry (:237-297) + backtrack(depth,top,bot,...) (:333-353) + finalize (:303-324).
 * Returns 1 iff done with this read.  Partials found are left in s->partials. */
static int dfs_backtrack(dfs_t* s, uint32_t ham)
{
	const bto_index* ix = s->ix;
	int nsInFtab = 0;
	s->npartials = 0;
	/* NB tallyNs indexes qry[qlen-i-1] for i < _3revOff: callers keep r3 <= qlen */
	if (!dfs_tally_ns(s, &nsInFtab)) return 0;
	uint32_t ftabChars = ix->ftabChars;
	uint32_t m = s->unrev < s->qlen ? s->unrev : s->qlen;
	int ret;
	s->bailed = 0;
	size_t need = (size_t)s->qlen * s->qlen;
	s->pairs = (uint32_t*)calloc(need * 8 + 8, 4);
	s->elims = (uint8_t*)calloc(need + 1, 1);
	if (nsInFtab == 0 && m >= ftabChars) {
		uint32_t ftabOff = s->qry[s->qlen - ftabChars];
		for (uint32_t i = ftabChars - 1; i > 0; i--) { ftabOff <<= 2; ftabOff |= s->qry[s->qlen - i]; }
		uint32_t top = bto_ftab_hi(ix, ftabOff), bot = bto_ftab_lo(ix, ftabOff + 1);
		if (s->cnt) s->cnt->ftab++;
		if (s->qlen == ftabChars && bot > top) {
			if (s->reportPartials > 0)
				ret = dfs_frame(s, 0, 0, s->unrev, s->r1, s->r2, s->r3, 0, 0, ham, ham, s->pairs, s->elims, nsInFtab > 0);
			else
				ret = dfs_report_alignment(s, 0, top, bot, (uint16_t)ham);
		} else if (bot > top) {
			ret = dfs_frame(s, 0, ftabChars, s->unrev, s->r1, s->r2, s->r3, top, bot, ham, ham, s->pairs, s->elims, nsInFtab > 0);
		} else ret = 0;
	} else {
		ret = dfs_frame(s, 0, 0, s->unrev, s->r1, s->r2, s->r3, 0, 0, ham, ham, s->pairs, s->elims, nsInFtab > 0);
	}
	free(s->pairs); free(s->elims); s->pairs = NULL; s->elims = NULL;
	s->numBts = 0;                      /* _totNumBts += _numBts; _numBts = 0 (:348-349) */
	if (s->reportPartials > 0 && s->npartials > 0) ret = 1;     /* finalize() */
	return ret;
}

/* PartialAlignmentManager::toMutsString (ebwt_search_util.h:310-362) + setMuts/applyPartialMutations
 * (ebwt_search_backtrack.h:146-157, 1368-1382).  seq/quals are the strings the extending searcher
 * uses, i.e. s->qry / s->qual (full length).  Returns oldQuals. */
static uint8_t dfs_apply_partial(dfs_t* s, uint64_t pal)
{
	uint32_t plen = s->fullLen;
	uint32_t pos[3] = { (uint32_t)(pal & 0xffff), (uint32_t)((pal >> 16) & 0xffff), (uint32_t)((pal >> 32) & 0xffff) };
	uint32_t chr[3] = { (uint32_t)((pal >> 48) & 3), (uint32_t)((pal >> 50) & 3), (uint32_t)((pal >> 52) & 3) };
	uint8_t oldQuals = 0;
	s->nmuts = 0;
	for (int k = 0; k < 3; k++) {
		if (k > 0 && pos[k] == 0xffff) break;
		uint32_t tpos = plen - 1 - pos[k];
		oldQuals = (uint8_t)(oldQuals + mm_penalty(s->maqPenalty, phred_of(s->qual[tpos])));
		s->muts[s->nmuts].pos = (uint16_t)tpos;
		s->muts[s->nmuts].oldBase = s->qry[tpos];
		s->muts[s->nmuts].newBase = (uint8_t)chr[k];
		s->nmuts++;
	}
	for (int k = 0; k < s->nmuts; k++) s->qry[s->muts[k].pos] = s->muts[k].newBase;
	return oldQuals;
}
static void dfs_undo_partial(dfs_t* s)
{
	for (int k = 0; k < s->nmuts; k++) s->qry[s->muts[k].pos] = s->muts[k].oldBase;
	s->nmuts = 0;
}

static void dfs_init(dfs_t* s, sink_t* sink, bt_op_counts* cnt, uint32_t qualThresh, uint32_t maxBts,
                     uint32_t reportPartials, int considerQuals, int halfAndHalf, int maqPenalty)
{
	memset(s, 0, sizeof(*s));
	s->sink = sink; s->cnt = cnt;
	s->qualThresh = qualThresh; s->maxBts = maxBts; s->reportPartials = reportPartials;
	s->reportExacts = 1; s->considerQuals = considerQuals; s->halfAndHalf = halfAndHalf; s->maqPenalty = maqPenalty;
}

/* ------------------------------------------------------------------------------------------
 * Worker-loop bodies = the phase scripts
 * ---------------------------------------------------------------------------------------- */
typedef struct {
	const bto_index *fwIx, *bwIx;
	const bt_policy* pol;
	const uint8_t *seq, *qual; uint32_t len, seed;
	sink_t* sink; bt_op_counts* cnt;
} rd_t;

#define SETQ(S, IX, FW) dfs_set_query((S), (IX), (FW), r->seq, r->qual, r->len, r->seed)

/* search_exact.c:9-26 (exactSearchWorker, ebwt_search.cpp:1130) */
static int run_v0(const rd_t* r)
{
	dfs_t* bt = (dfs_t*)malloc(sizeof(dfs_t));
	dfs_init(bt, r->sink, r->cnt, 0xffffffffu, 0xffffffffu, 0, 0, 0, 1);
	uint32_t plen = r->len;
	int done = 0;
	if (!r->pol->nofw) {
		SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, 0, plen, plen, plen, plen);
		done = dfs_backtrack(bt, 0);
	}
	if (!done && !r->pol->norc) {
		SETQ(bt, r->fwIx, 0); dfs_set_offs(bt, 0, 0, plen, plen, plen, plen);
		dfs_backtrack(bt, 0);
	}
	free(bt);
	return BT_OK;
}

/* search_1mm_phase1.c, search_1mm_phase2.c (mismatchSearchWorkerFull, ebwt_search.cpp:1606) */
static int run_v1(const rd_t* r)
{
	uint32_t plen = r->len, s = plen, s3 = s >> 1, s5 = (s >> 1) + (s & 1);
	int nofw = r->pol->nofw, norc = r->pol->norc;
	if (plen < 2) return BT_ERR_READ_SHORT;
	dfs_t* bt = (dfs_t*)malloc(sizeof(dfs_t));
	dfs_init(bt, r->sink, r->cnt, 0xffffffffu, 0xffffffffu, 0, 0, 0, 1);
	int done = 0;
	bt->reportExacts = 1;
	if (!done && !nofw) { SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, 0, s, s, s, s); done = dfs_backtrack(bt, 0); }
	if (!done && !norc) { SETQ(bt, r->fwIx, 0); dfs_set_offs(bt, 0, 0, s, s, s, s); done = dfs_backtrack(bt, 0); }
	bt->reportExacts = 0;
	if (!done && !norc) { SETQ(bt, r->fwIx, 0); dfs_set_offs(bt, 0, 0, s5, s, s, s); done = dfs_backtrack(bt, 0); }
	if (!done && !nofw) { SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, 0, s5, s, s, s); done = dfs_backtrack(bt, 0); }
	if (!done && !norc) { SETQ(bt, r->bwIx, 0); dfs_set_offs(bt, 0, 0, s3, s, s, s); done = dfs_backtrack(bt, 0); }
	if (!done && !nofw) { SETQ(bt, r->bwIx, 1); dfs_set_offs(bt, 0, 0, s3, s, s, s); done = dfs_backtrack(bt, 0); }
	free(bt);
	return BT_OK;
}

/* search_23mm_phase1-3.c with two=true (twoOrThreeMismatchSearchWorkerFull, ebwt_search.cpp:2056) */
static int run_v2(const rd_t* r)
{
	uint32_t plen = r->len, s = plen, s3 = s >> 1, s5 = (s >> 1) + (s & 1);
	int nofw = r->pol->nofw, norc = r->pol->norc;
	if (plen < 4) return BT_ERR_READ_SHORT;      /* search_23mm_phase1.c:13-20 */
	dfs_t* bt = (dfs_t*)malloc(sizeof(dfs_t));
	int done = 0;
	/* btr1 -> fw index */
	dfs_init(bt, r->sink, r->cnt, 0xffffffffu, 0xffffffffu, 0, 0, 0, 1);
	bt->reportExacts = 1;
	if (!done && !nofw) { SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, 0, plen, plen, plen, plen); done = dfs_backtrack(bt, 0); }
	if (!done && !norc) { SETQ(bt, r->fwIx, 0); dfs_set_offs(bt, 0, 0, s5, s5, s, s); done = dfs_backtrack(bt, 0); }
	/* bt2 -> mirror index */
	bt->reportExacts = 0;
	if (!done && !nofw) { SETQ(bt, r->bwIx, 1); dfs_set_offs(bt, 0, 0, s5, s5, s, s); done = dfs_backtrack(bt, 0); }
	if (!done && !norc) { SETQ(bt, r->bwIx, 0); dfs_set_offs(bt, 0, 0, s3, s3, s, s); done = dfs_backtrack(bt, 0); }
	/* bt3 -> fw index; bthh3 half-and-half */
	if (!done && !nofw) {
		SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, 0, s3, s3, s, s); done = dfs_backtrack(bt, 0);
		if (!done) {
			bt->halfAndHalf = 1; bt->reportExacts = 1;
			SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, s3, s, 0, s3, s, s); done = dfs_backtrack(bt, 0);
			bt->halfAndHalf = 0;
		}
	}
	if (!done && !norc) {
		bt->halfAndHalf = 1; bt->reportExacts = 1;
		SETQ(bt, r->fwIx, 0); dfs_set_offs(bt, s5, s, 0, s5, s, s); done = dfs_backtrack(bt, 0);
	}
	free(bt);
	return BT_OK;
}

/* search_seeded_phase1-4.c (seededQualSearchWorkerFull, ebwt_search.cpp:2378-2585) */
static int run_seeded(const rd_t* r, uint32_t* status)
{
	const bt_policy* pol = r->pol;
	int nofw = pol->nofw, norc = pol->norc, maq = pol->maq_round;
	uint32_t seedMms = (uint32_t)pol->mms, seedLen = (uint32_t)pol->seed_len;
	uint32_t qualCutoff = (uint32_t)pol->qual_thresh, maxBts = (uint32_t)pol->max_bts;
	uint32_t plen = r->len, s = seedLen, s3 = s >> 1, s5 = (s >> 1) + (s & 1);
	uint32_t qs = plen < s ? plen : s, qs3 = qs >> 1, qs5 = (qs >> 1) + (qs & 1);
	/* phase 1 prologue: skip short reads and reads with too many Ns in the seed */
	int skip = 0;
	if (plen < 4) skip = 1;
	else {
		uint32_t slen = plen < seedLen ? plen : seedLen; uint32_t ns = 0;
		for (uint32_t i = 0; i < slen; i++) if (r->seq[i] == 4 && ++ns > seedMms) { skip = 1; break; }
	}
	if (skip) { *status |= BT_ST_SKIPPED; return BT_OK; }
	/* the seed bounds the scripts pass when the read is shorter than the seed */
	uint32_t S = (qs < s) ? qs : s, S3 = (qs < s) ? qs3 : s3, S5 = (qs < s) ? qs5 : s5;
	dfs_t* bt = (dfs_t*)malloc(sizeof(dfs_t));
	uint64_t* pals = (uint64_t*)malloc(sizeof(uint64_t) * 4096);
	int npals = 0, done = 0;
	/* btf1: fw index, fw read, exact end-to-end, quals ignored */
	if (!nofw) {
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 0, 0, 1);
		SETQ(bt, r->fwIx, 1); dfs_set_offs(bt, 0, plen, plen, plen, plen, plen);
		done = dfs_backtrack(bt, 0);
	}
	/* bt1: fw index, rc read: cases 1R-3R */
	if (!done && !norc) {
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 0, maq);
		dfs_set_offs(bt, 0, 0, seedMms > 0 ? S5 : S, seedMms > 1 ? S5 : S, seedMms > 2 ? S5 : S, seedMms > 3 ? S5 : S);
		SETQ(bt, r->fwIx, 0);
		done = dfs_backtrack(bt, 0);
	}
	/* btf2: mirror index, fw read: cases 1F-3F, no exacts */
	if (!done && !nofw) {
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 0, maq);
		bt->reportExacts = 0;
		SETQ(bt, r->bwIx, 1);
		dfs_set_offs(bt, 0, 0, seedMms > 0 ? S5 : S, seedMms > 1 ? S5 : S, seedMms > 2 ? S5 : S, seedMms > 3 ? S5 : S);
		done = dfs_backtrack(bt, 0);
	}
	if (done || seedMms == 0) goto out;
	/* btr2: mirror index, rc read, seed only: seedlings for case 4R */
	if (!norc) {
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, seedMms, 1, 0, maq);
		bt->reportExacts = nofw ? 1 : 0;      /* btr2.setReportExacts(false) sits inside if(!nofw) */
		dfs_set_offs(bt, 0, 0, S3, seedMms > 1 ? S3 : S, seedMms > 2 ? S3 : S, seedMms > 3 ? S3 : S);
		SETQ(bt, r->bwIx, 0);
		bt->qlen = bt->fullLen < s ? bt->fullLen : s;              /* setQlen(s) */
		dfs_backtrack(bt, 0);
		npals = bt->npartials; memcpy(pals, bt->partials, sizeof(uint64_t) * (size_t)npals);
	}
	/* phase 3 */
	if (!norc) {
		/* btr3: fw index, rc read: extend the 4R seedlings (RNG seeded once, by setQuery) */
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 0, maq);
		bt->reportExacts = 1;
		SETQ(bt, r->fwIx, 0);
		if (npals > 0) {
			dfs_set_offs(bt, 0, 0, S, S, S, S);
			for (int i = 0; i < npals && !done; i++) {
				uint8_t oldQuals = dfs_apply_partial(bt, pals[i]);
				done = dfs_backtrack(bt, oldQuals);
				dfs_undo_partial(bt);
			}
		}
		if (done) goto out;
		if (seedMms >= 2) {
			/* btr23: half-and-half, fw index, rc read */
			dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 1, maq);
			SETQ(bt, r->fwIx, 0);
			dfs_set_offs(bt, S5, S, 0, (seedMms <= 2) ? S5 : 0, (seedMms < 3) ? S : S5, S);
			done = dfs_backtrack(bt, 0);
			if (done) goto out;
		}
	}
	if (nofw) goto out;
	/* btf3: fw index, fw read, seed only: seedlings for case 4F */
	dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, seedMms, 1, 0, maq);
	SETQ(bt, r->fwIx, 1);
	bt->qlen = bt->fullLen < seedLen ? bt->fullLen : seedLen;         /* setQlen(seedLen) */
	dfs_set_offs(bt, 0, 0, S3, seedMms > 1 ? S3 : S, seedMms > 2 ? S3 : S, seedMms > 3 ? S3 : S);
	dfs_backtrack(bt, 0);
	npals = bt->npartials; memcpy(pals, bt->partials, sizeof(uint64_t) * (size_t)npals);
	/* phase 4: btf4: mirror index, fw read: extend the 4F seedlings */
	dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 0, maq);
	bt->reportExacts = 1;
	SETQ(bt, r->bwIx, 1);
	if (npals > 0) {
		dfs_set_offs(bt, 0, 0, S, S, S, S);
		for (int i = 0; i < npals && !done; i++) {
			uint8_t oldQuals = dfs_apply_partial(bt, pals[i]);
			done = dfs_backtrack(bt, oldQuals);
			dfs_undo_partial(bt);
		}
	}
	if (done) goto out;
	if (seedMms >= 2) {
		/* btf24: half-and-half, mirror index, fw read */
		dfs_init(bt, r->sink, r->cnt, qualCutoff, maxBts, 0, 1, 1, maq);
		SETQ(bt, r->bwIx, 1);
		dfs_set_offs(bt, S5, S, 0, (seedMms <= 2) ? S5 : 0, (seedMms < 3) ? S : S5, S);
		done = dfs_backtrack(bt, 0);
	}
out:
	free(pals); free(bt);
	return BT_OK;
}

int bto_align_read(const bto_index* ixFw, const bto_index* ixBw, const bt_policy* pol,
                   const uint8_t* seq, const uint8_t* qual, int len, uint32_t seed,
                   bto_hit* hits, int cap, uint32_t* n_hits_total, uint32_t* status,
                   bt_op_counts* counts)
{
	if (len < 0 || len > BTO_MAXLEN) return -BT_ERR_ARG;
	if (len == 0 && !pol->best) {
		/* a read trimmed away (-3/-5): skipped by the seeded worker (search_seeded_phase1.c:17-44), an
		 * error for -v 1/2 (search_1mm_phase1.c:12-15, search_23mm_phase1.c:13-20), nothing to find for -v 0 */
		if (n_hits_total) *n_hits_total = 0;
		if (status) *status = pol->mode == BT_MODE_N ? BT_ST_SKIPPED : (pol->mms > 0 ? BT_ST_TOOSHORT : 0);
		return 0;
	}
	if (pol->best) return bto_align_read_best(ixFw, ixBw, pol, seq, qual, len, seed, hits, cap, n_hits_total, status, counts);
	sink_t sink;
	memset(&sink, 0, sizeof(sink));
	sink.n = pol->all_hits ? 0xffffffffu : pol->khits;
	sink.max = pol->mhits;
	sink.all = pol->all_hits;
	sink.hits = hits; sink.cap = cap;
	rd_t r = { ixFw, ixBw, pol, seq, qual, (uint32_t)len, seed, &sink, counts };
	uint32_t st = 0;
	int rc;
	if (pol->mode == BT_MODE_V) {
		if (pol->mms == 0) rc = run_v0(&r);
		else if (pol->mms == 1) rc = run_v1(&r);
		else if (pol->mms == 2) rc = run_v2(&r);
		else rc = BT_ERR_ARG;
	} else {
		rc = run_seeded(&r, &st);
	}
	if (rc != BT_OK) return -rc;
	if (sink.dropped && sink.hitsForThisRead <= sink.max) st |= BT_ST_HITCAP;
	if (n_hits_total) *n_hits_total = sink.hitsForThisRead;
	if (status) *status = st;
	/* finishRead (hit.h:741-786): maxed reads report nothing; otherwise keep the first _n */
	if (sink.hitsForThisRead > sink.max) return 0;
	int n = sink.stored;
	if ((uint32_t)n > sink.n) n = (int)sink.n;
	return n;
}
