/*
 * bt_host.cpp -- .ebwt loader and phase-program compiler (see bt_host.h).
 */
#include "bt_host.h"
#include <stdio.h>
#include <string.h>

namespace {
struct File {
	FILE* f = nullptr;
	explicit File(const std::string& p) { f = fopen(p.c_str(), "rb"); }
	~File() { if (f) fclose(f); }
	bool rd(void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
};
}  // namespace

int bt_host_index_load(const std::string& base, bool fw, int offrate_override, BtIndexHost* out)
{
	BtIndexHost& h = *out;
	h = BtIndexHost();
	h.fw = fw;
	File f1(base + ".1.ebwt");
	if (!f1.f) return BT_ERR_IO;
	uint32_t one = 0;
	if (!f1.rd(&one, 4)) return BT_ERR_IO;
	if (one != 1) return BT_ERR_FORMAT;                 /* other-endian or not an index */
	if (!f1.rd(&h.len, 4) || !f1.rd(&h.lineRate, 4) || !f1.rd(&h.linesPerSide, 4) ||
	    !f1.rd(&h.offRate, 4) || !f1.rd(&h.ftabChars, 4) || !f1.rd(&h.flags, 4)) return BT_ERR_IO;
	/* SideLocus::initFromRow hard-codes 224 symbols per side (ebwt.h:1477): only lineRate 6,
	 * linesPerSide 1 is a valid small index, and that is all bowtie-build emits. */
	if (h.lineRate != 6 || h.linesPerSide != 1 || h.ftabChars < 1 || h.ftabChars > 15 ||
	    h.offRate < 0 || h.offRate > 31) return BT_ERR_FORMAT;
	if (h.flags < 0 && ((-h.flags) & 4)) return BT_ERR_FORMAT;    /* EBWT_ENTIRE_REV: bt2 layout */
	const uint32_t bwtLen = h.len + 1u;
	const uint32_t bwtSz = h.len / 4u + 1u;
	const uint32_t numSidePairs = (bwtSz + 2u * 56u - 1u) / (2u * 56u);
	const uint64_t ebwtTotLen = (uint64_t)numSidePairs * 128u;
	const uint32_t ftabLen = (1u << (2 * h.ftabChars)) + 1u;
	const uint32_t eftabLen = 2u * (uint32_t)h.ftabChars;
	const uint32_t offsLen = (uint32_t)(((uint64_t)bwtLen + (1ull << h.offRate) - 1ull) >> h.offRate);
	if (!f1.rd(&h.nPat, 4)) return BT_ERR_IO;
	h.plen.resize(h.nPat);
	if (!f1.rd(h.plen.data(), 4ull * h.nPat) || !f1.rd(&h.nFrag, 4)) return BT_ERR_IO;
	h.rstarts.resize(3ull * h.nFrag);
	h.ebwt.resize(ebwtTotLen);
	h.ftab.resize(ftabLen);
	h.eftab.resize(eftabLen);
	if (!f1.rd(h.rstarts.data(), 12ull * h.nFrag) || !f1.rd(h.ebwt.data(), ebwtTotLen) ||
	    !f1.rd(&h.zOff, 4) || !f1.rd(h.fchr, 20) || !f1.rd(h.ftab.data(), 4ull * ftabLen) ||
	    !f1.rd(h.eftab.data(), 4ull * eftabLen)) return BT_ERR_IO;
	/* reference names: '\n'-separated, '\0'-terminated (ebwt.h:3452-3531) */
	{
		std::string cur;
		int c;
		while ((c = fgetc(f1.f)) != EOF && c != 0) {
			if (c == '\n') { if (h.refnames.size() < h.nPat) h.refnames.push_back(cur); cur.clear(); }
			else cur.push_back((char)c);
		}
		if (!cur.empty() && h.refnames.size() < h.nPat) h.refnames.push_back(cur);
		while (h.refnames.size() < h.nPat) h.refnames.push_back(std::to_string(h.refnames.size()));
	}
	File f2(base + ".2.ebwt");
	if (!f2.f) return BT_ERR_IO;
	if (!f2.rd(&one, 4)) return BT_ERR_IO;
	if (one != 1) return BT_ERR_FORMAT;
	std::vector<uint32_t> offs(offsLen);
	if (!f2.rd(offs.data(), 4ull * offsLen)) return BT_ERR_IO;
	if (offrate_override > h.offRate && offrate_override < 32) {
		const uint32_t diff = (uint32_t)(offrate_override - h.offRate);
		uint32_t sampled = offsLen >> diff;
		if ((offsLen & ~(0xffffffffu << diff)) != 0) sampled++;
		h.offs.resize(sampled);
		for (uint32_t i = 0, idx = 0; i < offsLen; i += (1u << diff)) h.offs[idx++] = offs[i];
		h.offRate = offrate_override;
	} else {
		h.offs.swap(offs);
	}
	return BT_OK;
}

void bt_host_index_describe(const BtIndexHost& h, BtIndexDev* d)
{
	memset(d, 0, sizeof(*d));
	d->len = h.len; d->zOff = h.zOff; d->ftabChars = (uint32_t)h.ftabChars;
	d->offRate = (uint32_t)h.offRate; d->offMask = 0xffffffffu << h.offRate;
	d->nFrag = h.nFrag; d->nPat = h.nPat; d->fw = h.fw ? 1u : 0u;
	for (int i = 0; i < 5; i++) d->fchr[i] = h.fchr[i];
	/* postReadInit (ebwt.h:1043-1059), restated as (side, storage-symbol) of '$' */
	d->zSide = h.zOff / 224u;
	const uint32_t co = h.zOff % 224u;
	d->zSym = (d->zSide & 1u) ? co : (223u - co);
}

void bt_host_restore_text(const BtIndexHost& h, uint8_t* out)
{
	BtIndexDev d;
	bt_host_index_describe(h, &d);
	std::vector<uint8_t> padded(h.ebwt);
	padded.resize(padded.size() + 128);
	d.ebwt = padded.data();
	uint32_t i = h.len, jumps = 0;            /* the row of the suffix "$" (sorts last) */
	while (i != h.zOff && jumps < h.len) {
		uint32_t lf[4], L;
		bt_rank4(d, i, lf, &L);
		out[h.len - 1u - jumps] = (uint8_t)L;
		i = lf[L];
		jumps++;
	}
}

/* ---- phase programs ------------------------------------------------------------------------ */
static BtStep mk(bool mirror, bool readFw, int kind, bool re, bool cq, bool hh, bool maq, int rp,
                 int o0, int o1, int o2, int o3, int o4, int o5, uint32_t qt, uint32_t mb)
{
	BtStep s;
	memset(&s, 0, sizeof(s));
	s.mirror = mirror; s.readFw = readFw; s.kind = (uint8_t)kind; s.reportExacts = re; s.considerQuals = cq;
	s.halfAndHalf = hh; s.maq = maq; s.reportPartials = (uint8_t)rp;
	s.oc[0] = (uint8_t)o0; s.oc[1] = (uint8_t)o1; s.oc[2] = (uint8_t)o2; s.oc[3] = (uint8_t)o3;
	s.oc[4] = (uint8_t)o4; s.oc[5] = (uint8_t)o5;
	s.qualThresh = qt; s.maxBts = mb;
	return s;
}

int bt_host_compile_program(const bt_policy& pol, BtProgram* prog)
{
	BtProgram& P = *prog;
	memset(&P, 0, sizeof(P));
	const bool nofw = pol.nofw != 0, norc = pol.norc != 0;
	const int Z = BT_OC_ZERO, PL = BT_OC_PLEN, S = BT_OC_S, S3 = BT_OC_S3, S5 = BT_OC_S5;
	const uint32_t INF = 0xffffffffu;
	std::vector<BtStep> st;
	P.sinkAll = pol.all_hits ? 1u : 0u;
	P.sinkN = pol.all_hits ? INF : pol.khits;
	P.sinkMax = pol.mhits;
	if (P.sinkN == 0) return BT_ERR_ARG;
	if (pol.mode == BT_MODE_V) {
		P.seeded = 0; P.seedLen = INF; P.seedMms = (uint32_t)pol.mms;
		if (pol.mms == 0) {
			/* search_exact.c:9-26 */
			P.minLen = 0;
			if (!nofw) st.push_back(mk(0, 1, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, PL, PL, PL, PL, INF, INF));
			if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, PL, PL, PL, PL, INF, INF));
		} else if (pol.mms == 1) {
			/* search_1mm_phase1.c:17-66, search_1mm_phase2.c:12-30 */
			P.minLen = 2;
			if (!nofw) st.push_back(mk(0, 1, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, S, S, S, S, INF, INF));
			if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, S, S, S, S, INF, INF));
			if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S5, S, S, S, INF, INF));
			if (!nofw) st.push_back(mk(0, 1, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S5, S, S, S, INF, INF));
			if (!norc) st.push_back(mk(1, 0, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S3, S, S, S, INF, INF));
			if (!nofw) st.push_back(mk(1, 1, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S3, S, S, S, INF, INF));
		} else if (pol.mms == 2) {
			/* search_23mm_phase1.c:21-42, phase2.c:11-37, phase3.c:9-66 (two = true) */
			P.minLen = 4;
			if (!nofw) st.push_back(mk(0, 1, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, PL, PL, PL, PL, INF, INF));
			if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, Z, S5, S5, S, S, INF, INF));
			if (!nofw) st.push_back(mk(1, 1, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S5, S5, S, S, INF, INF));
			if (!norc) st.push_back(mk(1, 0, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S3, S3, S, S, INF, INF));
			if (!nofw) {
				st.push_back(mk(0, 1, BT_KIND_SEARCH, 0, 0, 0, 1, 0, Z, Z, S3, S3, S, S, INF, INF));
				st.push_back(mk(0, 1, BT_KIND_SEARCH, 1, 0, 1, 1, 0, S3, S, Z, S3, S, S, INF, INF));
			}
			if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 0, 1, 1, 0, S5, S, Z, S5, S, S, INF, INF));
		} else {
			return BT_ERR_ARG;     /* -v 3 runs the best-first engine in the reference */
		}
	} else if (pol.mode == BT_MODE_N) {
		/* search_seeded_phase1.c:46-79, phase2.c:9-72, phase3.c:9-121, phase4.c:9-91 */
		if (pol.mms < 0 || pol.mms > 3 || pol.seed_len < 5) return BT_ERR_ARG;
		const int n = pol.mms;
		const bool maq = pol.maq_round != 0;
		const uint32_t qt = (uint32_t)pol.qual_thresh, mb = (uint32_t)pol.max_bts;
		P.seeded = 1; P.seedLen = (uint32_t)pol.seed_len; P.seedMms = (uint32_t)n; P.minLen = 0;
		const int a0 = n > 0 ? S5 : S, a1 = n > 1 ? S5 : S, a2 = n > 2 ? S5 : S, a3 = n > 3 ? S5 : S;
		const int b1 = n > 1 ? S3 : S, b2 = n > 2 ? S3 : S, b3 = n > 3 ? S3 : S;
		if (!nofw) st.push_back(mk(0, 1, BT_KIND_SEARCH, 1, 0, 0, 1, 0, Z, PL, PL, PL, PL, PL, qt, mb));
		if (!norc) st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 1, 0, maq, 0, Z, Z, a0, a1, a2, a3, qt, mb));
		if (!nofw) st.push_back(mk(1, 1, BT_KIND_SEARCH, 0, 1, 0, maq, 0, Z, Z, a0, a1, a2, a3, qt, mb));
		if (n > 0) {
			if (!norc) {
				st.push_back(mk(1, 0, BT_KIND_GEN, nofw ? 1 : 0, 1, 0, maq, n, Z, Z, S3, b1, b2, b3, qt, mb));
				st.push_back(mk(0, 0, BT_KIND_EXTEND, 1, 1, 0, maq, 0, Z, Z, S, S, S, S, qt, mb));
				if (n >= 2)
					st.push_back(mk(0, 0, BT_KIND_SEARCH, 1, 1, 1, maq, 0, S5, S, Z, (n <= 2) ? S5 : Z, (n < 3) ? S : S5, S, qt, mb));
			}
			if (!nofw) {
				st.push_back(mk(0, 1, BT_KIND_GEN, 1, 1, 0, maq, n, Z, Z, S3, b1, b2, b3, qt, mb));
				st.push_back(mk(1, 1, BT_KIND_EXTEND, 1, 1, 0, maq, 0, Z, Z, S, S, S, S, qt, mb));
				if (n >= 2)
					st.push_back(mk(1, 1, BT_KIND_SEARCH, 1, 1, 1, maq, 0, S5, S, Z, (n <= 2) ? S5 : Z, (n < 3) ? S : S5, S, qt, mb));
			}
		}
	} else {
		return BT_ERR_ARG;
	}
	if (st.size() > BT_MAX_STEPS) return BT_ERR_ARG;
	P.nsteps = (int32_t)st.size();
	for (size_t i = 0; i < st.size(); i++) P.steps[i] = st[i];
	return BT_OK;
}
