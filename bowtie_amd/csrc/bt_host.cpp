/*
 * bt_host.cpp -- .ebwt loader and phase-program compiler (see bt_host.h).
 */
#include "bt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if BT_WIDE
#include <thread>
#endif

namespace {
struct File {
	FILE* f = nullptr;
	explicit File(const std::string& p) { f = fopen(p.c_str(), "rb"); }
	~File() { if (f) fclose(f); }
	bool rd(void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
};
}  // namespace

/* ---- the .ebwt family -----------------------------------------------------------------------------
 * Four on-disk variants hold the same index (ebwt.h:2835-3445, EbwtParams::init ebwt.h:138-183):
 *
 *   <base>.1.ebwt   32-bit offsets, 64-byte sides in fw/bw pairs: 56 BWT bytes + 2 u32 counters
 *                   ([A][C] after a backward side, [G][T] after a forward one), 224 rows per side
 *   <base>.1.ebwtl  the BOWTIE_64BIT_INDEX build (btypes.h:4-14): every offset is a u64, lineRate 7,
 *                   128-byte sides = 112 BWT bytes + 2 u64 counters, 448 rows per side
 *   <base>.1.bt2    bowtie2-build's layout (isBt2Index, ebwt.h:173-180,2240-2328): all sides forward,
 *   <base>.1.bt2l   sideSz - 4*OFF_SIZE BWT bytes followed by the 4 counters [A][C][G][T] before the side
 *
 * each in either byte order (first word 1 or 1<<24, ebwt.h:2926-2937).  Whatever the file holds, the
 * image the 32-bit build's kernels read is the first layout: the BWT symbols are re-dealt into 224-row
 * sides and the counters recomputed, offsets are narrowed to 32 bits (an index of 2^32-1 or more rows
 * is BT_ERR_ROWS64 there: its rows are u32).  What does not survive the conversion, and so is kept as a
 * flag, is the 64-bit build's arithmetic: h.wide (see BtIndexDev::wide).  The build with 64-bit rows
 * (-DBT_WIDE=1, bt_rank.h "the row type") keeps every offset 64 bits wide and derives its rank blocks
 * straight from the file's BWT (build_blocks below).                                                 */
namespace {
struct Reader {
	File& f; bool swap, wide, ok = true, narrow = true;
	uint32_t u32() {
		uint32_t v = 0;
		if (!f.rd(&v, 4)) ok = false;
		return swap ? __builtin_bswap32(v) : v;
	}
	uint64_t off() {                    /* a TIndexOffU */
		if (!wide) return u32();
		uint64_t v = 0;
		if (!f.rd(&v, 8)) ok = false;
		return swap ? __builtin_bswap64(v) : v;
	}
	/* n offsets -> rows of this build's width.  Narrow build: eftab references in ftab are ~k in the file's width, the low
	 * word is ~k.  Wide build from a 32-bit file: ~k of 32 bits becomes ~k of 64 (`codes`: the array is ftab) */
	bool offs(std::vector<bt_row>& out, uint64_t n, bool check, bool codes = false, uint64_t lenForCodes = 0) {
		out.resize((size_t)n);
		if (!wide) {
#if BT_WIDE
			std::vector<uint32_t> buf((size_t)(n < (1u << 20) ? n : (1u << 20)));
			for (uint64_t i = 0; i < n; ) {
				const uint64_t m = n - i < buf.size() ? n - i : buf.size();
				if (!f.rd(buf.data(), 4ull * m)) return ok = false;
				for (uint64_t k = 0; k < m; k++) {
					const uint32_t v = swap ? __builtin_bswap32(buf[(size_t)k]) : buf[(size_t)k];
					out[(size_t)(i + k)] = (codes && v > lenForCodes) ? (0xffffffff00000000ull | v) : (uint64_t)v;
				}
				i += m;
			}
			(void)check;
			return true;
#else
			(void)codes; (void)lenForCodes;
			if (!f.rd(out.data(), 4ull * n)) return ok = false;
			if (swap) for (auto& v : out) v = __builtin_bswap32(v);
			return true;
#endif
		}
		std::vector<uint64_t> buf((size_t)(n < (1u << 20) ? n : (1u << 20)));
		for (uint64_t i = 0; i < n; ) {
			const uint64_t m = n - i < buf.size() ? n - i : buf.size();
			if (!f.rd(buf.data(), 8ull * m)) return ok = false;
			for (uint64_t k = 0; k < m; k++) {
				const uint64_t v = swap ? __builtin_bswap64(buf[(size_t)k]) : buf[(size_t)k];
#if BT_WIDE
				out[(size_t)(i + k)] = v;
#else
				if (check && (v >> 32) != 0 && (v >> 32) != 0xffffffffull) narrow = false;
				out[(size_t)(i + k)] = (uint32_t)v;
#endif
			}
			i += m;
		}
		return true;
	}
};

/* Source BWT -> the 64-byte-side layout.  sym(row) of the source: */
struct SrcBwt {
	const uint8_t* p; uint32_t sideSz, sideBwtSz; bool bt2;
	uint32_t sym(uint64_t row) const {
		const uint32_t per = sideBwtSz * 4u;
		const uint64_t side = row / per; uint32_t co = (uint32_t)(row % per);
		uint32_t by = co >> 2, bp = co & 3u;
		if (!bt2 && (side & 1u) == 0) { by = sideBwtSz - by - 1u; bp ^= 3u; }    /* backward sides run the other way */
		return (p[side * sideSz + by] >> (2u * bp)) & 3u;
	}
};

#if BT_WIDE
/* The wide build's rank blocks (bt_rank.h), straight from the file's BWT: rows 0 .. len, '$' (row zOff, stored as an A) not
 * counted.  One pass over the BWT makes the bit planes and per-chunk counts (threads), a prefix sum over the chunks the
 * segment table, a second pass over the blocks their counters, relative to their segment's start.  fchrB: fchr + rowBias. */
void build_blocks(const SrcBwt& src, uint64_t len, uint64_t zOff, const bt_row fchrB[5], uint32_t segShift,
                  std::vector<uint8_t>* blk, std::vector<uint64_t>* segBase)
{
	const uint64_t nb = bt_blk_count(len), nRows = len + 1u;
	blk->assign((size_t)(nb * BT_BLK_BYTES), 0);
	const uint32_t chunkShift = segShift < 14u ? segShift : 14u;          /* blocks per chunk: at most a segment */
	const uint64_t nChunks = (nb + (1ull << chunkShift) - 1u) >> chunkShift;
	std::vector<uint64_t> cc((size_t)nChunks * 4u, 0);                      /* symbol counts per chunk */
	unsigned nt = std::thread::hardware_concurrency();
	if (const char* e = getenv("BT_LOAD_THREADS")) nt = (unsigned)atoi(e);
	if (nt < 1) nt = 1;
	if (nt > 64) nt = 64;
	if ((uint64_t)nt > nChunks) nt = (unsigned)nChunks;
	auto planes = [&](unsigned t) {
		for (uint64_t ch = t; ch < nChunks; ch += nt) {
			uint64_t cnt[4] = {0, 0, 0, 0};
			const uint64_t b0 = ch << chunkShift, b1 = (b0 + (1ull << chunkShift)) < nb ? b0 + (1ull << chunkShift) : nb;
			for (uint64_t b = b0; b < b1; b++) {
				uint64_t p0 = 0, p1 = 0;
				for (uint32_t i = 0; i < BT_BLK_ROWS; i++) {
					const uint64_t row = b * BT_BLK_ROWS + i;
					if (row >= nRows) break;
					const uint32_t c = src.sym(row);
					p0 |= (uint64_t)(c & 1u) << i; p1 |= (uint64_t)(c >> 1) << i;
					if (row != zOff) cnt[c]++;
				}
				uint32_t w[4] = {(uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32)};
				memcpy(blk->data() + b * BT_BLK_BYTES + 16, w, 16);
			}
			for (int c = 0; c < 4; c++) cc[(size_t)ch * 4u + c] = cnt[c];
		}
	};
	{
		std::vector<std::thread> th;
		for (unsigned t = 1; t < nt; t++) th.emplace_back(planes, t);
		planes(0);
		for (auto& x : th) x.join();
	}
	/* chunk starts (absolute counts) and the segment table */
	const uint64_t nSeg = ((nb - 1u) >> segShift) + 1u;
	segBase->assign((size_t)nSeg * 4u, 0);
	std::vector<uint64_t> start((size_t)nChunks * 4u, 0);
	{
		uint64_t run[4] = {0, 0, 0, 0};
		for (uint64_t ch = 0; ch < nChunks; ch++) {
			for (int c = 0; c < 4; c++) start[(size_t)ch * 4u + c] = run[c];
			if (((ch << chunkShift) & ((1ull << segShift) - 1u)) == 0)
				for (int c = 0; c < 4; c++) (*segBase)[(size_t)((ch << chunkShift) >> segShift) * 4u + c] = fchrB[c] + run[c];
			for (int c = 0; c < 4; c++) run[c] += cc[(size_t)ch * 4u + c];
		}
	}
	auto counters = [&](unsigned t) {
		for (uint64_t ch = t; ch < nChunks; ch += nt) {
			const uint64_t b0 = ch << chunkShift, b1 = (b0 + (1ull << chunkShift)) < nb ? b0 + (1ull << chunkShift) : nb;
			const uint64_t seg = b0 >> segShift;
			uint64_t run[4];
			for (int c = 0; c < 4; c++) run[c] = fchrB[c] + start[(size_t)ch * 4u + c] - (*segBase)[(size_t)seg * 4u + c];
			for (uint64_t b = b0; b < b1; b++) {
				uint32_t w[4] = {(uint32_t)run[0], (uint32_t)run[1], (uint32_t)run[2], (uint32_t)run[3]};
				memcpy(blk->data() + b * BT_BLK_BYTES, w, 16);
				uint32_t pw[4];
				memcpy(pw, blk->data() + b * BT_BLK_BYTES + 16, 16);
				const uint64_t p0 = ((uint64_t)pw[1] << 32) | pw[0], p1 = ((uint64_t)pw[3] << 32) | pw[2];
				const uint64_t first = b * BT_BLK_ROWS;
				const uint32_t valid = first >= nRows ? 0u : (nRows - first >= BT_BLK_ROWS ? BT_BLK_ROWS : (uint32_t)(nRows - first));
				const uint64_t m = valid >= 64u ? ~0ull : ((1ull << valid) - 1ull);
				const uint64_t cT = (uint64_t)__builtin_popcountll(p0 & p1 & m), cLo = (uint64_t)__builtin_popcountll(p0 & m), cHi = (uint64_t)__builtin_popcountll(p1 & m);
				uint64_t cA = valid - cLo - cHi + cT;
				if (zOff >= first && zOff < first + valid) cA--;
				run[0] += cA; run[1] += cLo - cT; run[2] += cHi - cT; run[3] += cT;
			}
		}
	};
	{
		std::vector<std::thread> th;
		for (unsigned t = 1; t < nt; t++) th.emplace_back(counters, t);
		counters(0);
		for (auto& x : th) x.join();
	}
}
#else
void repack_sides(const SrcBwt& src, uint64_t srcRows, uint32_t zOff, uint32_t len, std::vector<uint8_t>* out)
{
	const uint32_t bwtSz = len / 4u + 1u;
	const uint32_t numSidePairs = (bwtSz + 2u * 56u - 1u) / (2u * 56u);
	out->assign((size_t)numSidePairs * 128u, 0);
	uint32_t cnt[4] = {0, 0, 0, 0};                  /* occurrences in rows before the cursor, '$' not counted */
	uint64_t row = 0;
	for (uint32_t p = 0; p < numSidePairs; p++) {
		uint8_t* bw = out->data() + (size_t)p * 128u; uint8_t* fwd = bw + 64;
		for (uint32_t k = 0; k < 224u; k++, row++) {          /* backward side: stored reversed */
			const uint32_t c = row < srcRows ? src.sym(row) : 0u;
			if (row != zOff) cnt[c]++;
			const uint32_t by = 56u - (k >> 2) - 1u, bp = (k & 3u) ^ 3u;
			bw[by] |= (uint8_t)(c << (2u * bp));
		}
		/* after the backward side: [A][C] here, [G][T] behind the forward side -- all four as of this point */
		memcpy(bw + 56, &cnt[0], 4); memcpy(bw + 60, &cnt[1], 4);
		memcpy(fwd + 56, &cnt[2], 4); memcpy(fwd + 60, &cnt[3], 4);
		for (uint32_t k = 0; k < 224u; k++, row++) {
			const uint32_t c = row < srcRows ? src.sym(row) : 0u;
			if (row != zOff) cnt[c]++;
			fwd[k >> 2] |= (uint8_t)(c << (2u * (k & 3u)));
		}
	}
}

#endif /* BT_WIDE */

const char* const kExt[4] = {"bt2", "ebwt", "bt2l", "ebwtl"};
}  // namespace

/* Which of the four an index base names: bowtie prefers .bt2 over .ebwt (adjustEbwtBase, ebwt.cpp:36-48)
 * and its wrapper picks the 64-bit binary only when there is no small index (bowtie:52-81). */
int bt_host_index_variant(const std::string& base)
{
	/* BT_INDEX_PREFER_LARGE (set by bowtie-amd --large-index): the 64-bit files ONLY, as bowtie-align-l -- the binary the
	 * reference's wrapper then runs (bowtie:64-65) -- knows no others: a base that has only .ebwt / .bt2 files is "not found" */
	static const int plain[4] = {0, 1, 2, 3}, large[4] = {2, 3, 2, 3};
	const char* e = getenv("BT_INDEX_PREFER_LARGE");
	const int* order = (e && *e && *e != '0') ? large : plain;
	for (int k = 0; k < 4; k++) {
		File f(base + ".1." + kExt[order[k]]);
		if (f.f) return order[k];
	}
	return -1;
}
const char* bt_host_index_ext(int variant) { return variant >= 0 && variant < 4 ? kExt[variant] : "ebwt"; }

/* The text length in the header of <base>.1.<ext>, whichever variant the base names: a few bytes read, nothing loaded.
 * (bowtie-amd decides with it which binary runs -- before a single read has been taken from the input; the reference's
 * wrapper looks at the file names for the same decision, bowtie:52-81.) */
int bt_host_index_header_len(const std::string& base, uint64_t* len64)
{
	const int variant = bt_host_index_variant(base);
	if (variant < 0) return BT_ERR_IO;
	File f1(base + ".1." + kExt[variant]);
	if (!f1.f) return BT_ERR_IO;
	uint32_t one = 0;
	if (!f1.rd(&one, 4)) return BT_ERR_IO;
	if (one != 1 && one != (1u << 24)) return BT_ERR_FORMAT;
	Reader R{f1, one != 1, variant >= 2};
	*len64 = R.off();
	return R.ok ? BT_OK : BT_ERR_IO;
}

int bt_host_index_load(const std::string& base, bool fw, int offrate_override, BtIndexHost* out, int variant)
{
	BtIndexHost& h = *out;
	h = BtIndexHost();
	h.fw = fw;
	if (variant < 0) variant = bt_host_index_variant(base);
	if (variant < 0) return BT_ERR_IO;
	const bool bt2 = (variant & 1) == 0, wide = variant >= 2;
	const std::string ext = kExt[variant];
	File f1(base + ".1." + ext);
	if (!f1.f) return BT_ERR_IO;
	uint32_t one = 0;
	if (!f1.rd(&one, 4)) return BT_ERR_IO;
	if (one != 1 && one != (1u << 24)) return BT_ERR_FORMAT;          /* not an index */
	Reader R{f1, one != 1, wide};
	const uint64_t len64 = R.off();
	h.lineRate = (int32_t)R.u32(); h.linesPerSide = (int32_t)R.u32(); h.offRate = (int32_t)R.u32();
	h.ftabChars = (int32_t)R.u32(); h.flags = (int32_t)R.u32();
	if (!R.ok) return BT_ERR_IO;
	/* SideLocus::initFromRow hard-codes the rows per side (ebwt.h:1469-1479): 56*OFF_SIZE*4 (48*OFF_SIZE*4 for
	 * bt2), i.e. lineRate 6 (7 when wide) with one line per side is the only geometry that works there */
	const int32_t wantLine = wide ? 7 : 6;
	if (bt2) h.linesPerSide = 1;                                       /* EbwtParams::init ebwt.h:149 */
	if (h.lineRate != wantLine || h.linesPerSide != 1 || h.ftabChars < 1 || h.ftabChars > 15 ||
	    h.offRate < 0 || h.offRate > 31) return BT_ERR_FORMAT;
#if BT_WIDE
	if (len64 == 0 || len64 >= (1ull << 38)) return BT_ERR_FORMAT;     /* block numbers are 32-bit */
#else
	if (len64 == 0) return BT_ERR_FORMAT;
	if (len64 >= 0xffffffffull) return BT_ERR_ROWS64;                  /* this build's rows are 32-bit: libbowtie_amd_l.so */
#endif
	h.len = (bt_row)len64;
	h.wide = wide; h.bt2 = bt2; h.swapped = R.swap;
	const uint32_t offSize = wide ? 8u : 4u;
	const uint32_t srcSideSz = 1u << h.lineRate;
	const uint32_t srcSideBwtSz = srcSideSz - (bt2 ? 4u : 2u) * offSize;
	const uint64_t bwtLen = len64 + 1u;
	const uint64_t bwtSz = len64 / 4u + 1u;
	uint64_t srcTotLen;
	if (bt2) srcTotLen = (uint64_t)((bwtSz + srcSideBwtSz - 1u) / srcSideBwtSz) * srcSideSz;
	else     srcTotLen = (uint64_t)((bwtSz + 2u * srcSideBwtSz - 1u) / (2u * srcSideBwtSz)) * (2u * srcSideSz);
	const uint32_t ftabLen = (1u << (2 * h.ftabChars)) + 1u;
	const uint32_t eftabLen = 2u * (uint32_t)h.ftabChars;
	const uint64_t offsLen = (bwtLen + (1ull << h.offRate) - 1ull) >> h.offRate;
	const uint64_t nPat = R.off();
	if (!R.ok) return BT_ERR_IO;
	if (nPat == 0 || nPat > h.len) return BT_ERR_FORMAT;
	h.nPat = (uint32_t)nPat;
	if (!R.offs(h.plen, nPat, true)) return BT_ERR_IO;
#if BT_WIDE
	/* an alignment's offset within its sequence is 32 bits in bt_hit (and in the reference's Hit::h.second only because
	 * TIndexOffU is 64 there): a single sequence of 2^32 bases or more is not held */
	for (const bt_row v : h.plen) if (v >= 0xffffffffull) return BT_ERR_FORMAT;
#endif
	const uint64_t nFrag = R.off();
	if (!R.ok) return BT_ERR_IO;
	if (nFrag == 0 || nFrag > h.len) return BT_ERR_FORMAT;
	h.nFrag = (uint32_t)nFrag;
	if (!R.offs(h.rstarts, 3ull * nFrag, true)) return BT_ERR_IO;
	std::vector<uint8_t> src;
#if BT_WIDE
	std::vector<uint8_t>& raw = src;
#else
	std::vector<uint8_t>& raw = (!wide && !bt2) ? h.ebwt : src;
#endif
	raw.resize((size_t)srcTotLen);
	if (!f1.rd(raw.data(), (size_t)srcTotLen)) return BT_ERR_IO;
	if (R.swap && !wide && !bt2) {
		/* the two counters behind each side are words too (ebwt.h:3154-3162); the other layouts' counters
		 * are recomputed below and never read */
		for (uint64_t o = 56; o + 8 <= srcTotLen; o += 64) {
			uint32_t c[2];
			memcpy(c, raw.data() + o, 8);
			c[0] = __builtin_bswap32(c[0]); c[1] = __builtin_bswap32(c[1]);
			memcpy(raw.data() + o, c, 8);
		}
	}
	const uint64_t zOff = R.off();
	uint64_t fchr[5];
	for (int i = 0; i < 5; i++) fchr[i] = R.off();
	if (!R.ok) return BT_ERR_IO;
	if (zOff > h.len || fchr[4] != h.len) return BT_ERR_FORMAT;
	for (int i = 0; i < 5; i++) { if (fchr[i] > h.len || (i && fchr[i] < fchr[i - 1])) return BT_ERR_FORMAT; h.fchr[i] = (bt_row)fchr[i]; }
	h.zOff = (bt_row)zOff;
	if (!R.offs(h.ftab, ftabLen, false, true, len64) || !R.offs(h.eftab, eftabLen, true)) return BT_ERR_IO;
	if (!R.narrow) return BT_ERR_FORMAT;
#if BT_WIDE
	{
		/* the test knobs (bt_host.h): rows numbered from a bias, small segments -- how rows beyond 2^32 are exercised on a
		 * genome of a few Mbp.  Honoured only where BT_TEST_KNOBS=1 says this is a test (tests/conftest.py sets it) */
		const char* tk = getenv("BT_TEST_KNOBS");
		const bool knobs = tk && *tk && *tk != '0';
		if (const char* e = knobs ? getenv("BT_WIDE_SEG_SHIFT") : nullptr) { const long v = atol(e); if (v >= 2 && v <= 25) h.segShift = (uint32_t)v; }
		if (const char* e = knobs ? getenv("BT_WIDE_ROW_BIAS") : nullptr) {
			const uint64_t b = strtoull(e, nullptr, 0);
			/* whole segments, whole 16-byte pieces of the SA sample (two entries), block numbers that stay 32 bits */
			if ((b & ((1ull << (h.segShift + 6u)) - 1u)) != 0 || (b & ((2ull << h.offRate) - 1u)) != 0 || b >= (1ull << 37) || b + len64 >= (1ull << 38)) return BT_ERR_ARG;
			h.rowBias = b;
		}
		const uint64_t B = h.rowBias;
		for (int i = 0; i < 5; i++) h.fchr[i] += B;
		h.zOff += B;
		for (auto& v : h.ftab) if (v <= len64) v += B;
		for (auto& v : h.eftab) v += B;
		SrcBwt sb{src.data(), srcSideSz, srcSideBwtSz, bt2};
		build_blocks(sb, len64, zOff, h.fchr, h.segShift, &h.blk, &h.segBase);
		std::vector<uint8_t>().swap(src);
		h.lineRate = 6;
	}
#else
	if (wide || bt2) {
		SrcBwt sb{src.data(), srcSideSz, srcSideBwtSz, bt2};
		repack_sides(sb, bwtLen, h.zOff, h.len, &h.ebwt);
		h.lineRate = 6;
	}
#endif
	/* fragments must tile the joined text in order (joinedToTextOff's binary search, ebwt.h:2569-2629) */
	for (uint32_t i = 0; i < h.nFrag; i++) {
		const bt_row lo = h.rstarts[3 * i], up = i + 1 < h.nFrag ? h.rstarts[3 * i + 3] : h.len;
		if (lo >= up || up > h.len || h.rstarts[3 * i + 1] >= h.nPat) return BT_ERR_FORMAT;
	}
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
	File f2(base + ".2." + ext);
	if (!f2.f) return BT_ERR_IO;
	if (!f2.rd(&one, 4)) return BT_ERR_IO;
	if (one != (R.swap ? (1u << 24) : 1u)) return BT_ERR_FORMAT;
	Reader R2{f2, R.swap, wide};
	std::vector<bt_row> offs;
	if (!R2.offs(offs, offsLen, true)) return BT_ERR_IO;
	if (!R2.narrow) return BT_ERR_FORMAT;
	if (offrate_override > h.offRate && offrate_override < 32) {
		const uint32_t diff = (uint32_t)(offrate_override - h.offRate);
		uint64_t sampled = offsLen >> diff;
		if ((offsLen & ~(~0ull << diff)) != 0) sampled++;
		h.offs.resize((size_t)sampled);
		for (uint64_t i = 0, idx = 0; i < offsLen; i += (1ull << diff)) h.offs[(size_t)idx++] = offs[(size_t)i];
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
	d->offRate = (uint32_t)h.offRate; d->offMask = (bt_row)BT_OFF_MASK << h.offRate;
	d->nFrag = h.nFrag; d->nPat = h.nPat; d->fw = h.fw ? 1u : 0u; d->wide = h.wide ? 1u : 0u;
	for (int i = 0; i < 5; i++) d->fchr[i] = h.fchr[i];
	/* postReadInit (ebwt.h:1043-1059), restated as (side, storage-symbol) of '$' */
	d->zSide = (uint32_t)(h.zOff / 224u);
	const uint32_t co = (uint32_t)(h.zOff % 224u);
	d->zSym = (d->zSide & 1u) ? co : (223u - co);
	d->zBlk = (uint32_t)(h.zOff / BT_BLK_ROWS); d->zPos = (uint32_t)(h.zOff % BT_BLK_ROWS);
#if BT_WIDE
	d->segShift = h.segShift; d->rowLim = h.len + h.rowBias;
#endif
}

void bt_host_restore_text(const BtIndexHost& h, uint8_t* out)
{
	BtIndexDev d;
	bt_host_index_describe(h, &d);
#if BT_WIDE
	d.blk = h.blk.data(); d.segBase = h.segBase.data();
	bt_host_index_bias(h, &d);
	bt_row i = d.rowLim, jumps = 0;           /* the row of the suffix "$" (sorts last) */
	while (i != h.zOff && jumps < h.len) {
		bt_row lf[4]; uint32_t L;
		bt_rank4(d, i, lf, &L);
		out[h.len - 1u - jumps] = (uint8_t)L;
		i = lf[L];
		jumps++;
	}
}
#else
	std::vector<uint8_t> padded(h.ebwt);
	padded.resize(padded.size() + 128);
	d.ebwt = padded.data();
	uint32_t i = h.len, jumps = 0;            /* the row of the suffix "$" (sorts last) */
	while (i != h.zOff && jumps < h.len) {
		uint32_t lf[4], L;
		bt_rank4_sides(d, i, lf, &L);
		out[h.len - 1u - jumps] = (uint8_t)L;
		i = lf[L];
		jumps++;
	}
}
#endif

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

/* ---- best-first driver trees ------------------------------------------------------------------
 * The tree of RangeSourceDrivers the reference's Unpaired*AlignerFactory::create() builds for a
 * policy (aligner_0mm.h:69-115, aligner_1mm.h:73-152, aligner_23mm.h:73-236,
 * aligner_seed_mm.h:82-516), as data for the device automaton of bt_best.h. */
namespace {
struct TreeBuilder {
	BfProgram& P;
	int add_spec(bool mirror, bool fw, uint32_t qualLim, bool reportExacts, int hh, bool partial, bool seed,
	             uint32_t seedLen, bool nudgeLeft, int r0, int r1, int r2, int r3, bool useBtCnt)
	{
		if (P.nspecs >= BF_MAX_SPECS) return -1;
		BfSpec& s = P.specs[P.nspecs];
		memset(&s, 0, sizeof(s));
		s.mirror = mirror; s.fw = fw; s.reportExacts = reportExacts; s.halfAndHalf = (uint8_t)hh; s.partial = partial;
		s.seed = seed; s.nudgeLeft = nudgeLeft; s.useBtCnt = useBtCnt;
		s.rev[0] = (uint8_t)r0; s.rev[1] = (uint8_t)r1; s.rev[2] = (uint8_t)r2; s.rev[3] = (uint8_t)r3;
		s.qualLim = qualLim; s.seedLen = seedLen;
		if (mirror) P.needMirror = 1;
		return (int)P.nspecs++;
	}
	bool leaf(int spec)
	{
		if (spec < 0 || P.nnodes >= BF_MAX_NODES) return false;
		BfNode& n = P.nodes[P.nnodes++];
		n.kind = BF_LEAF; n.spec = (uint8_t)spec; n.genSpec = 0; n.fw = P.specs[spec].fw;
		return true;
	}
	bool seeded(int factSpec, int genSpec, bool fw)
	{
		if (factSpec < 0 || genSpec < 0 || P.nnodes >= BF_MAX_NODES) return false;
		BfNode& n = P.nodes[P.nnodes++];
		n.kind = BF_SEEDED; n.spec = (uint8_t)factSpec; n.genSpec = (uint8_t)genSpec; n.fw = fw;
		return true;
	}
};
}

/* the drivers of one (mate, strand) block, in the order the factories push them.  `paired` selects
 * the Paired*AlignerFactory variants, which differ from the unpaired ones in two places: the
 * nudgeLeft flags of -v 1 (aligner_1mm.h:295-408) and rev1Off of the -v 3 half-and-half driver
 * (aligner_23mm.h:403-411, 469-477, 534-542, 598-606). */
static bool add_block(TreeBuilder& T, const bt_policy& pol, bool fw, int mate, bool paired)
{
	const uint32_t INF = 0xffffffffu;
	const int B = BF_PIN_BEGIN, L = BF_PIN_LEN, H = BF_PIN_HI_HALF, S = BF_PIN_SEED;
	bool ok = true;
	/* fw read: mirror index first; rc read: text index first */
	const bool m1 = fw, m2 = !fw;
	auto spec = [&](bool mirror, uint32_t qualLim, bool exact, int hh, bool partial, bool seed, uint32_t sl, bool nudge,
	                int r0, int r1, int r2, int r3, bool bc) {
		const int id = T.add_spec(mirror, fw, qualLim, exact, hh, partial, seed, sl, nudge, r0, r1, r2, r3, bc);
		if (id >= 0) T.P.specs[id].mate = (uint8_t)mate;
		return id;
	};
	if (pol.mode == BT_MODE_V) {
		if (pol.mms == 0) {
			ok = ok && T.leaf(spec(false, INF, true, 0, false, false, 0, true, L, L, L, L, false));
		} else if (pol.mms == 1) {
			ok = ok && T.leaf(spec(m1, INF, true, 0, false, false, 0, paired ? true : !m1, H, L, L, L, false));
			ok = ok && T.leaf(spec(m2, INF, false, 0, false, false, 0, paired ? false : !m2, H, L, L, L, false));
		} else {
			const bool two = pol.mms == 2;
			const int r2 = two ? L : H;
			ok = ok && T.leaf(spec(m1, INF, true, 0, false, false, 0, true, H, H, r2, L, false));
			ok = ok && T.leaf(spec(m2, INF, false, 0, false, false, 0, false, H, H, r2, L, false));
			ok = ok && T.leaf(spec(m1, INF, false, 2, false, false, 0, true, B, H, r2, L, false));
			if (!two) ok = ok && T.leaf(spec(m2, INF, false, 3, false, false, 0, false, B, (paired && !(mate == 0 && !fw)) ? B : H, H, L, false));
		}
	} else {
		const uint32_t q = (uint32_t)pol.qual_thresh, sl = (uint32_t)pol.seed_len;
		const int n = pol.mms;
		const bool bc = n >= 2;          /* the backtrack budget only exists for -n 2/3 (aligner_seed_mm.h:99,134) */
		if (n == 0) {
			ok = ok && T.leaf(spec(m1, q, true, 0, false, false, sl, true, S, S, S, S, false));
		} else {
			const int a1 = n >= 2 ? H : S, a2 = n >= 3 ? H : S;
			ok = ok && T.leaf(spec(m1, q, true, 0, false, false, sl, true, H, a1, a2, S, bc));
			{
				const int f = spec(m1, q, true, 0, false, false, sl, true, S, S, S, S, bc);
				ok = ok && T.seeded(f, spec(m2, q, false, 0, true, true, sl, false, H, a1, a2, S, bc), fw);
			}
			if (n >= 3) {
				const int f = spec(m1, q, true, 0, false, false, sl, true, S, S, S, S, bc);
				ok = ok && T.seeded(f, spec(m2, q, false, 3, true, true, sl, false, B, H, H, S, bc), fw);
			}
			if (n >= 2) ok = ok && T.leaf(spec(m1, q, false, 2, false, false, sl, true, B, H, a2, S, bc));
		}
	}
	return ok;
}

static int compile_best(const bt_policy& pol, bool paired, BfProgram* prog)
{
	BfProgram& P = *prog;
	memset(&P, 0, sizeof(P));
	if (pol.mms < 0 || pol.mms > 3) return BT_ERR_ARG;
	if (pol.mode != BT_MODE_V && pol.mode != BT_MODE_N) return BT_ERR_ARG;
	if (pol.mode == BT_MODE_N && pol.seed_len < 5) return BT_ERR_ARG;
	const uint32_t INF = 0xffffffffu;
	const uint32_t mult = paired ? 2u : 1u;       /* createMult(2): mates count separately (hit.h:1012-1016, 1155-1159, 1246-1249) */
	P.maq = pol.maq_round ? 1u : 0u;
	P.maxBts = (uint32_t)pol.max_bts;
	P.strandFix = 1;                                   /* ebwt_search.cpp:227 */
	P.btCntOn = (pol.mode == BT_MODE_N && pol.mms >= 2) ? 1u : 0u;
	/* createSinkFactory (ebwt_search.cpp:992-1020) */
	P.sinkStrata = pol.strata ? 1u : 0u;
	P.sinkAll = pol.all_hits ? 1u : 0u;
	P.sinkN = pol.all_hits ? (pol.strata ? (INF / 2u) * mult : INF) : pol.khits * mult;
	P.sinkMax = pol.mhits == INF ? INF : pol.mhits * mult;
	P.sampleMax = pol.sample_max ? 1u : 0u;
	if (P.sinkN == 0) return BT_ERR_ARG;
	TreeBuilder T{P};
	bool ok = true;
	if (!paired) {
		if (!pol.nofw) ok = ok && add_block(T, pol, true, 0, false);
		if (!pol.norc) ok = ok && add_block(T, pol, false, 0, false);
	} else {
		/* Paired*AlignerFactory::create() with v1_ == false: -v: 1Fw 1Rc 2Fw 2Rc (aligner_0mm.h:320-327,
		 * aligner_1mm.h:286-415, aligner_23mm.h:358-606); -n: 1Fw 2Fw 1Rc 2Rc (aligner_seed_mm.h:705-1290) */
		if (pol.max_ins < 0 || pol.min_ins < 0 || pol.pair_tries < 0) return BT_ERR_ARG;
		bool d1f = true, d1r = true, d2f = true, d2r = true;
		if (pol.nofw) { if (pol.mate1_fw) d1f = false; else d1r = false; if (pol.mate2_fw) d2f = false; else d2r = false; }
		if (pol.norc) { if (pol.mate1_fw) d1r = false; else d1f = false; if (pol.mate2_fw) d2r = false; else d2f = false; }
		const int ov[4][2] = {{0, 1}, {0, 0}, {1, 1}, {1, 0}}, on[4][2] = {{0, 1}, {1, 1}, {0, 0}, {1, 0}};
		for (int k = 0; k < 4; k++) {
			const int mate = pol.mode == BT_MODE_V ? ov[k][0] : on[k][0];
			const bool fw = (pol.mode == BT_MODE_V ? ov[k][1] : on[k][1]) != 0;
			const bool doit = mate == 0 ? (fw ? d1f : d1r) : (fw ? d2f : d2r);
			if (doit) ok = ok && add_block(T, pol, fw, mate, true);
		}
		P.paired = 1; P.minIns = (uint32_t)pol.min_ins; P.maxIns = (uint32_t)pol.max_ins;
		P.mate1Fw = pol.mate1_fw ? 1u : 0u; P.mate2Fw = pol.mate2_fw ? 1u : 0u;
		P.pairTries = (uint32_t)pol.pair_tries; P.allowContain = pol.allow_contain ? 1u : 0u;
		/* without --best the same drivers are driven by PairedBWAlignerV1, each (mate, strand) block behind a
		 * cost-aware driver of its own ("if(v1_)" in the factories); symCeiling is -m (ebwt_search.cpp:1275) */
		if (pol.pe_v1) P.paired = 2;
		/* Exact/OneMM/TwoMM/ThreeMMRefAligner for -v, Seed{0..3}RefAligner(seedLen, qualCutoff) for -n
		 * (aligner_0mm.h:303, aligner_1mm.h:417, aligner_23mm.h:608-612, aligner_seed_mm.h:668-676) */
		P.refSeeded = pol.mode == BT_MODE_N ? 1u : 0u; P.refMms = (uint32_t)pol.mms;
		P.refSeedLen = (uint32_t)pol.seed_len; P.refQualMax = pol.mode == BT_MODE_N ? (uint32_t)pol.qual_thresh : INF;
	}
	return ok ? BT_OK : BT_ERR_ARG;
}

int bt_host_compile_best(const bt_policy& pol, BfProgram* prog) { return compile_best(pol, false, prog); }
int bt_host_compile_best_paired(const bt_policy& pol, BfProgram* prog) { return compile_best(pol, true, prog); }

/* ---- the 2-bit reference ------------------------------------------------------------------------
 * <base>.3.ebwt: u32 1, u32 nRecords, then per unambiguous stretch { u32 off (Ns before it), u32 len,
 * u8 first (1 = first stretch of a sequence) }; <base>.4.ebwt: the bases of all stretches, 4 per byte,
 * first base in the low bits (reference.h:35-240, ref_read.h:57-87).  A sequence whose first record
 * has len 0 is all gaps and has no index in the Ebwt (reference.h:160-176). */
int bt_host_ref_load(const std::string& base, const BtIndexHost& idx, BtRefHost* out, int variant)
{
	if (variant < 0) variant = bt_host_index_variant(base);
	if (variant < 0) return BT_ERR_IO;
	const std::string ext = kExt[variant];
	File f3(base + ".3." + ext);
	if (!f3.f) return BT_ERR_IO;
	uint32_t one = 0;
	if (!f3.rd(&one, 4)) return BT_ERR_IO;
	if (one != 1 && one != (1u << 24)) return BT_ERR_FORMAT;
	Reader R3{f3, one != 1, variant >= 2};
	const uint64_t nrec64 = R3.off();
	if (!R3.ok) return BT_ERR_IO;
	if (nrec64 == 0 || nrec64 > (1u << 28)) return BT_ERR_FORMAT;
	const uint32_t nrec = (uint32_t)nrec64;
	struct Rec { uint32_t off, len; uint8_t first; };
	std::vector<Rec> recs(nrec);
	uint64_t cum = 0;
	for (uint32_t i = 0; i < nrec; i++) {
		const uint64_t o = R3.off(), l = R3.off();
		if (!R3.ok || !f3.rd(&recs[i].first, 1)) return BT_ERR_IO;
		if ((o >> 32) || (l >> 32)) return BT_ERR_FORMAT;
		recs[i].off = (uint32_t)o; recs[i].len = (uint32_t)l;
		cum += recs[i].len;
	}
	if (cum > idx.len) return BT_ERR_FORMAT;
	std::vector<uint8_t> packed((cum + 3) / 4);
	{
		File f4(base + ".4." + ext);
		if (!f4.f) return BT_ERR_IO;
		if (!f4.rd(packed.data(), packed.size())) return BT_ERR_IO;
	}
	BtRefHost& R = *out;
	R = BtRefHost();
	const uint32_t nRefs = idx.nPat;
	R.start.resize(nRefs); R.approxLen.assign(nRefs, 0);
	uint64_t total = 0;
	for (uint32_t t = 0; t < nRefs; t++) { R.start[t] = total; total += ((uint64_t)idx.plen[t] + 63u) & ~63ull; }
	R.bits.assign((size_t)(total / 16u) + 16u, 0u);            /* + padding: the mate finder reads 64-bit pieces past the end */
	R.nmask.assign((size_t)(total / 32u) + 16u, 0xffffffffu);
	int64_t t = -1; bool live = false; uint64_t pos = 0, src = 0;
	for (uint32_t i = 0; i < nrec; i++) {
		const Rec& r = recs[i];
		if (r.first) {
			live = r.len > 0;
			if (live) { t++; pos = 0; if ((uint64_t)t >= nRefs) return BT_ERR_FORMAT; }
		}
		if (r.len == 0) continue;
		if (!live) { src += r.len; continue; }
		pos += r.off;
		if (pos + r.len > idx.plen[(size_t)t]) return BT_ERR_FORMAT;
		const uint64_t g0 = R.start[(size_t)t] + pos;
		for (uint32_t k = 0; k < r.len; k++, src++) {
			const uint32_t c = (packed[(size_t)(src >> 2)] >> (2u * (uint32_t)(src & 3u))) & 3u;
			const uint64_t g = g0 + k;
			R.bits[(size_t)(g >> 4)] |= c << (2u * (uint32_t)(g & 15u));
			R.nmask[(size_t)(g >> 5)] &= ~(1u << (uint32_t)(g & 31u));
		}
		pos += r.len;
		R.approxLen[(size_t)t] = (uint32_t)pos;
	}
	if ((uint64_t)(t + 1) != nRefs) return BT_ERR_FORMAT;
	return BT_OK;
}
