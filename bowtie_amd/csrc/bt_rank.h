/*
 * bt_rank.h -- Occ/rank over one .ebwt side, the arithmetic core of LF-mapping.
 *
 * Behaviour follows Ebwt::countFwSideEx / countBwSideEx / mapLFEx / rowL of the reference
 * (ebwt.h:2081-2129, 2184-2226, 2334-2380, 1696-1704), restated in one unified form:
 *
 *   row  -> side = row / 224, charOff = row % 224, fw side iff side is odd
 *   a side stores 224 two-bit symbols in 56 bytes; symbol k of the *storage order* sits at
 *   bit 2k of the little-endian byte string.  A forward side stores BWT order; a backward side
 *   stores it reversed, so
 *        fw side: the rank query counts the first n = charOff        storage symbols
 *        bw side: the rank query counts the first n = 224 - charOff  storage symbols
 *   and LF(row,c) = fchr[c] + occ_mid[c] (+|-) cnt_n[c], where occ_mid are the four cumulative
 *   counters of the 128-byte side pair (A,C in bytes 56..63, G,T in bytes 120..127) valid at
 *   the pair's midpoint.  The '$' symbol is stored as an A; it is un-counted when it lies
 *   within the counted span (storage symbol zSym of side zSide, span test: n > zSym).
 *
 * That is the layout of the index FILES (bt_rank4_sides).  What the search kernels query is a second image the
 * loader derives from it once, laid out for this GPU rather than for a CPU's cache lines -- HBM is plentiful here
 * (288 GB; the image costs 0.5 byte per BWT row) and instruction issue is what the search is short of:
 *
 *   rank block b (32 bytes, one aligned sector) covers BWT rows 64b .. 64b+63:
 *       u32 occ[4]   LF(64b, c) for c = A,C,G,T  (absolute: fchr[c] + occurrences before the block, '$' excluded)
 *       u64 plane0   bit i = low  bit of the symbol at row 64b+i
 *       u64 plane1   bit i = high bit of the symbol at row 64b+i
 *   LF(row,c) = occ[c] + (number of c among the first row%64 symbols of the block): three 64-bit popcounts of
 *   masked planes (T = p0&p1, C = p0-T, G = p1-T, A by subtraction) -- about 35 instructions per row where the
 *   224-symbol side costs about 240, one 32-byte load instead of 64 + 8 bytes out of a 128-byte pair, no division
 *   by 224, no forward/backward sides.  Same numbers: the probe tests compare both with the reference's (App. D).
 *
 * Compiles for gfx950 (hipcc) and for the host (g++; used by the state-machine unit tests).
 */
#ifndef BT_RANK_H_
#define BT_RANK_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BT_HD __host__ __device__ __forceinline__
#else
#define BT_HD static inline
#endif
#if defined(__clang__)
#define BT_UNROLL _Pragma("unroll")
#define BT_NOUNROLL _Pragma("nounroll")
#else
#define BT_UNROLL
#define BT_NOUNROLL
#endif

struct alignas(16) BtU4 { uint32_t x, y, z, w; };

/* ---- the row type ---------------------------------------------------------------------------------------------------
 * The reference compiles its sources twice: bowtie-align-s with 32-bit offsets and, with -DBOWTIE_64BIT_INDEX, bowtie-align-l
 * with TIndexOffU = uint64_t for indexes of 2^32 - 1 rows and more (btypes.h:4-28; the `bowtie` wrapper picks the binary,
 * bowtie:52-81).  So does this tree: -DBT_WIDE=1 builds libbowtie_amd_l.so / bowtie-amd-l from the same sources with 64-bit
 * BWT rows and text offsets.  What changes with the width:
 *   - a row-valued table entry (ftab, eftab, offs, rstarts, plen, fchr) is 8 bytes, a 16-byte piece holds two of them;
 *   - a range-stack entry (tops ACGT, bots ACGT) is 64 bytes = four pieces, and a rank answer's quartets take two pieces each
 *     (BtRes, bt_core.h);
 *   - a rank block keeps its 32 bytes: its four counters are relative to the start of the block's SEGMENT (2^segShift blocks,
 *     2^31 rows by default), whose absolute counts sit in a small table (segBase);
 *   - a record of the best-first engine's arena (a branch, an alternative, a leaf's current range: bt_best.h) holds a row in
 *     two 32-bit words;
 *   - the wide build searches in row space only (no locus image: it would need a 64-bit form of its own).              */
#ifndef BT_WIDE
#define BT_WIDE 0
#endif
#if BT_WIDE
typedef uint64_t bt_row;
#define BT_OFF_MASK 0xffffffffffffffffull
#define BT_PIECE_ROWS 2u       /* row-valued entries per 16-byte piece */
#else
typedef uint32_t bt_row;
#define BT_OFF_MASK 0xffffffffu
#define BT_PIECE_ROWS 4u
#endif
#define BT_SEG_SHIFT_DEFAULT 25u   /* blocks per segment, log2: 2^25 blocks of 64 rows = 2^31 rows */
#define BT_SIDE_SYMS 224u

/* device-visible image of one index (fw or mirror); all pointers are device pointers on the
 * GPU build, host pointers in the host unit-test build. */
struct BtIndexDev {
	const uint8_t*  ebwt;      /* numSidePairs * 128 bytes, reference byte layout (NULL in the wide build: its loader
	                              derives the rank blocks straight from the file's BWT)                              */
	const bt_row* ftab;
	const bt_row* eftab;
	const bt_row* offs;
	const bt_row* rstarts;     /* 3 * nFrag                                                  */
	const bt_row* plen;
	bt_row   len, zOff;
	uint32_t zSide, zSym, ftabChars, offRate;
	bt_row   offMask;
	uint32_t nFrag, fw, nPat;
	bt_row   fchr[5];
	const uint8_t*  blk;       /* the rank blocks: (len + 1) / 64 + 2 of them, 32 bytes each (see above)          */
	uint32_t zBlk, zPos;       /* zOff / 64, zOff % 64                                                             */
	/* ---- the locus image (round 5; optional: loc == NULL means it was not built and the search stays in row space) ----
	 * Once a range is ONE BWT row the search is no longer a search: the row's suffix sits at one place of the text, and what
	 * the reference's mapLF1 / single-row mapLFEx steps compute base by base from there on is a string comparison against
	 * the text to the left of it.  HBM capacity (288 GB) buys that comparison:
	 *   loc[row]  16 bytes: word 0 = SA[row] (offset of the row's suffix in the joined text); words 1..3 = the 48 text
	 *             characters to the LEFT of it, two bits each, nearest first (depth order: character k = T[SA-1-k])
	 *   rtxt      the joined text REVERSED, two bits per base, 16 per word: base y = T[len-1-y] -- reversed so that the
	 *             characters further left of a position are at increasing addresses, like loc's words
	 *   walk[p]   u16, by text offset p: the LF steps Ebwt::reportChaseOne's walk (ebwt.h:2727-2746) takes from the row
	 *             whose suffix starts at p -- what the op counters tally for a hit that no longer walks (saturating) */
	const BtU4*     loc;
	const uint32_t* rtxt;
	const uint16_t* walk;
	/* ---- the jump table (round 6; optional: jump == NULL) ----
	 * The reference starts a search with one look-up for the query's first ftabChars characters (10 in bowtie-build's indexes)
	 * and then narrows the range a character at a time; on a 3 Gbp genome the next six of those steps go by before the range
	 * is one row, each of them a dependent rank -- a lock-step round -- and while the positions are ones the search may not
	 * revisit (the phase's `unrev` prefix: 14 characters with -n 2 -l 28, 28 for a seedling's extension, the whole read for
	 * the exact phase) nothing is chosen and nothing recorded there: the result is a function of the characters alone.  HBM
	 * buys the table of that function for the first jumpChars (14) characters: the range, and -- so that the op counters still
	 * say what the reference's algorithm does -- how many LF steps lie behind it, how many of them on two rows and how many of
	 * those inside one side pair.
	 *   jump[x]      two rows: top, bot (top == bot: the range became empty on the way)
	 *   jumpMeta[x]  u16: steps taken | two-row steps << 3 | same-pair steps << 6
	 * x = the characters in the order the search meets them, two bits each, first character lowest: its low 2 * ftabChars bits
	 * are the ftab offset. */
	const uint32_t* jump;
	const uint16_t* jumpMeta;
	uint32_t jumpChars, padJ;
	uint32_t wide;             /* the index FILES are a 64-bit (.ebwtl / .bt2l) build -- whatever this build's row type: the
	                              32-bit build holds such an index too if it has fewer than 2^32 - 1 rows.  The
	                              reference binary that serves it is compiled with 64-bit offsets, and two things
	                              a user can see follow the offset width: the row a hit is reported from is drawn
	                              with nextU<TIndexOffU>() (two draws, ebwt_search_backtrack.h:1538), and a
	                              best-first Branch is 160 bytes, so 1638 of them fit a pool chunk (pool.h:32)   */
#if BT_WIDE
	const uint64_t* segBase;   /* [segment][4]: LF(first row of the segment, ACGT), absolute                              */
	uint32_t segShift, padW;   /* blocks per segment, log2                                                                */
	bt_row   rowLim;           /* the last BWT row (= len, unless the image's rows are offset: tests, bt_host.h)           */
#endif
};
#if BT_WIDE
#define BT_ROWLIM(ix) ((ix).rowLim)
#else
#define BT_ROWLIM(ix) ((ix).len)
#endif

/* ---- global-memory accessors ---------------------------------------------------------------
 * Pointers that reach the device code through LDS or through structures in memory are generic,
 * and a load or store through a generic pointer is a FLAT instruction: it counts on the LDS
 * counter as well, so the next wait for an LDS read also waits for the HBM round trip of every
 * such access still in flight.  Everything the search kernel touches outside LDS is global
 * memory; BT_GP says so (no-op on the host build). */
#if defined(__HIPCC__)
typedef uint32_t bt_vec4 __attribute__((ext_vector_type(4)));
typedef uint32_t bt_vec2 __attribute__((ext_vector_type(2)));
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_GP(T, p) ((__attribute__((address_space(1))) T*)(p))
#else
#define BT_GP(T, p) ((T*)(p))
#endif
BT_HD BtU4 bt_ld4(const void* p)
{
	BtU4 r;
#if defined(__HIP_DEVICE_COMPILE__)
	const bt_vec4 v = *BT_GP(const bt_vec4, p);
	r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
#else
	memcpy(&r, p, 16);
#endif
	return r;
}
/* the same from an address that is only word-aligned (a window of the reversed text) */
BT_HD BtU4 bt_ld4w(const void* p)
{
	BtU4 r;
#if defined(__HIP_DEVICE_COMPILE__)
	typedef uint32_t bt_vec4w __attribute__((ext_vector_type(4), aligned(4)));
	const bt_vec4w v = *BT_GP(const bt_vec4w, p);
	r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
#else
	memcpy(&r, p, 16);
#endif
	return r;
}
BT_HD void bt_st4(void* p, const BtU4& v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	bt_vec4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
	*BT_GP(bt_vec4, p) = t;
#else
	memcpy(p, &v, 16);
#endif
}

/* counts of C,G,T among the first `bits` (0..32) symbols of w; A = bits - (C+G+T) */
BT_HD void bt_count_word(uint64_t w, uint32_t bits, uint32_t& cC, uint32_t& cG, uint32_t& cT)
{
	const uint64_t EVEN = 0x5555555555555555ull;
	uint64_t m = (bits >= 32) ? EVEN : (((1ull << (2 * bits)) - 1ull) & EVEN);
	uint64_t lo = w & m;
	uint64_t hi = (w >> 1) & m;
	cT += (uint32_t)__builtin_popcountll(hi & lo);
	cG += (uint32_t)__builtin_popcountll(hi & ~lo);
	cC += (uint32_t)__builtin_popcountll(~hi & lo);
}

#if !BT_WIDE
/* Rank from the 7 BWT words of the side holding `row` plus the pair's four counters.
 *   w[0..6] : the side's 56 BWT bytes as little-endian u64
 *   occ[4]  : occ_mid A,C,G,T of the side pair
 * -> lf[c] = LF(row,c) for c in ACGT, *L = BWT char at row (rowL). */
BT_HD void bt_rank4_words(const BtIndexDev& ix, uint32_t sideNum, uint32_t charOff,
                          const uint64_t w[7], const uint32_t occ[4], uint32_t lf[4], uint32_t* L)
{
	const bool fw = (sideNum & 1u) != 0;
	const uint32_t n  = fw ? charOff : (BT_SIDE_SYMS - charOff);
	const uint32_t li = fw ? charOff : (BT_SIDE_SYMS - 1u - charOff);
	uint32_t cC = 0, cG = 0, cT = 0;
BT_UNROLL
	for (int k = 0; k < 7; k++) {
		uint32_t lo = 32u * (uint32_t)k;
		uint32_t bits = n > lo ? n - lo : 0u;
		bt_count_word(w[k], bits, cC, cG, cT);
	}
	uint32_t cA = n - cC - cG - cT;
	if (sideNum == ix.zSide && n > ix.zSym) cA--;
	/* rowL: storage symbol li */
	uint64_t wl = w[0];
BT_UNROLL
	for (int k = 1; k < 7; k++) wl = ((li >> 5) == (uint32_t)k) ? w[k] : wl;
	*L = (uint32_t)(wl >> (2u * (li & 31u))) & 3u;
	if (fw) {
		lf[0] = ix.fchr[0] + occ[0] + cA;
		lf[1] = ix.fchr[1] + occ[1] + cC;
		lf[2] = ix.fchr[2] + occ[2] + cG;
		lf[3] = ix.fchr[3] + occ[3] + cT;
	} else {
		lf[0] = ix.fchr[0] + occ[0] - cA;
		lf[1] = ix.fchr[1] + occ[1] - cC;
		lf[2] = ix.fchr[2] + occ[2] - cG;
		lf[3] = ix.fchr[3] + occ[3] - cT;
	}
}

/* Rank straight from the index files' side layout: one lane fetches its own side (64 B) + the partner side's counters
 * (8 B).  Used to derive the rank blocks and by the probes; the search kernels use bt_rank4 below. */
BT_HD void bt_rank4_sides(const BtIndexDev& ix, uint32_t row, uint32_t lf[4], uint32_t* L)
{
	const uint32_t sideNum = row / BT_SIDE_SYMS;
	const uint32_t charOff = row - sideNum * BT_SIDE_SYMS;
	const uint64_t* side = (const uint64_t*)(ix.ebwt + (uint64_t)sideNum * 64u);
	uint64_t w[7];
BT_UNROLL
	for (int k = 0; k < 7; k++) w[k] = side[k];
	const uint64_t own = side[7];
	/* partner counters: fw side (odd) needs A,C from the end of the previous side; bw side
	 * (even) needs G,T from the end of the next side */
	const uint64_t oth = (sideNum & 1u) ? *(side - 1) : *(side + 15);
	uint32_t occ[4];
	if (sideNum & 1u) { occ[0] = (uint32_t)oth; occ[1] = (uint32_t)(oth >> 32); occ[2] = (uint32_t)own; occ[3] = (uint32_t)(own >> 32); }
	else              { occ[0] = (uint32_t)own; occ[1] = (uint32_t)(own >> 32); occ[2] = (uint32_t)oth; occ[3] = (uint32_t)(oth >> 32); }
	bt_rank4_words(ix, sideNum, charOff, w, occ, lf, L);
}
#endif /* !BT_WIDE */

/* ---- rank blocks ------------------------------------------------------------------------------------------------ */
#define BT_BLK_BYTES 32u
#define BT_BLK_ROWS 64u
BT_HD uint64_t bt_blk_count(bt_row len) { return ((uint64_t)len + 1u) / BT_BLK_ROWS + 2u; }
/* LF(row, ACGT) and rowL from a rank block's eight words: o = occ[4], p = plane0 lo/hi, plane1 lo/hi; n = row % 64;
 * zHere: the block is the one holding the '$' row (which is stored as an A and must not count as one) */
BT_HD void bt_rank4_blk(const BtU4& o, const BtU4& p, uint32_t n, bool zHere, uint32_t zPos, uint32_t lf[4], uint32_t* L)
{
	const uint64_t p0 = ((uint64_t)p.y << 32) | p.x, p1 = ((uint64_t)p.w << 32) | p.z;
	const uint64_t m = (1ull << n) - 1ull;                /* n = 0..63 */
	const uint64_t a = p0 & m, b = p1 & m;
	const uint32_t cLo = (uint32_t)__builtin_popcountll(a), cHi = (uint32_t)__builtin_popcountll(b), cT = (uint32_t)__builtin_popcountll(a & b);
	uint32_t cA = n - cLo - cHi + cT;
	if (zHere && n > zPos) cA--;
	lf[0] = o.x + cA; lf[1] = o.y + (cLo - cT); lf[2] = o.z + (cHi - cT); lf[3] = o.w + cT;
	*L = ((uint32_t)(p0 >> n) & 1u) | (((uint32_t)(p1 >> n) & 1u) << 1);
}
#define BT_LOC_CTX 48u
#define BT_RTXT_PAD_WORDS 16u       /* (the locus image, below) words of padding before rtxt[0] and after its last word */
BT_HD uint64_t bt_rtxt_words(bt_row len) { return ((uint64_t)len + 15u) / 16u + 2u * BT_RTXT_PAD_WORDS; }
/* the rank the search uses: one 32-byte block per BWT row queried */
#if BT_WIDE
/* the wide build's: the block's counters are relative to its segment's (see "the row type") */
BT_HD void bt_rank4(const BtIndexDev& ix, bt_row row, bt_row lf[4], uint32_t* L)
{
	const uint64_t b = row / BT_BLK_ROWS;
	const uint8_t* q = ix.blk + b * BT_BLK_BYTES;
	const BtU4 o = bt_ld4(q), p = bt_ld4(q + 16);
	uint32_t r[4];
	bt_rank4_blk(o, p, (uint32_t)(row % BT_BLK_ROWS), (uint32_t)b == ix.zBlk, ix.zPos, r, L);
	const uint64_t* sb = ix.segBase + (b >> ix.segShift) * 4u;
	for (int c = 0; c < 4; c++) lf[c] = BT_GP(const uint64_t, sb)[c] + r[c];
}
#else
BT_HD void bt_rank4(const BtIndexDev& ix, uint32_t row, uint32_t lf[4], uint32_t* L)
{
	const uint32_t b = row / BT_BLK_ROWS;
	const uint8_t* q = ix.blk + (uint64_t)b * BT_BLK_BYTES;
	const BtU4 o = bt_ld4(q), p = bt_ld4(q + 16);
	bt_rank4_blk(o, p, row % BT_BLK_ROWS, b == ix.zBlk, ix.zPos, lf, L);
}
/* Derive the rank blocks of one index from its side layout (host; the GPU loader does the same in a kernel,
 * bt_kernels.hip: bt_blk_build_kernel): running counts over the BWT, '$' skipped. */
BT_HD void bt_blk_build_host(const BtIndexDev& ix, uint8_t* out)
{
	uint32_t cnt[4] = {0, 0, 0, 0};
	const uint64_t nb = bt_blk_count(ix.len);
	for (uint64_t b = 0; b < nb; b++) {
		uint32_t w[8];
		for (int c = 0; c < 4; c++) w[c] = ix.fchr[c] + cnt[c];
		uint64_t p0 = 0, p1 = 0;
		for (uint32_t i = 0; i < BT_BLK_ROWS; i++) {
			const uint64_t row = b * BT_BLK_ROWS + i;
			if (row > ix.len) break;
			uint32_t lf[4], L;
			{
				/* rowL (ebwt.h:1696-1704): storage symbol charOff of a forward side, 223 - charOff of a backward one */
				const uint32_t sideNum = (uint32_t)row / BT_SIDE_SYMS, charOff = (uint32_t)row - sideNum * BT_SIDE_SYMS;
				const uint32_t li = (sideNum & 1u) ? charOff : (BT_SIDE_SYMS - 1u - charOff);
				L = (ix.ebwt[(uint64_t)sideNum * 64u + (li >> 2)] >> (2u * (li & 3u))) & 3u;
				(void)lf;
			}
			p0 |= (uint64_t)(L & 1u) << i; p1 |= (uint64_t)(L >> 1) << i;
			if (row != ix.zOff) cnt[L]++;
		}
		w[4] = (uint32_t)p0; w[5] = (uint32_t)(p0 >> 32); w[6] = (uint32_t)p1; w[7] = (uint32_t)(p1 >> 32);
		memcpy(out + b * BT_BLK_BYTES, w, BT_BLK_BYTES);
	}
}

/* ---- the locus image, host build (the emulator's; the GPU loader does the same with kernels, bt_kernels.hip) --------------
 * One pass over the text from its end: row 0 is the suffix "$" (SA = len); LF of the row of suffix p is the row of suffix
 * p - 1 and the BWT character there is T[p-1].  rtxt: (len + 15) / 16 + 32 words (padding: a window fetch may run past
 * either end), walk: len + 1 entries, loc: len + 1 records. */
BT_HD void bt_loc_build_host(const BtIndexDev& ix, BtU4* loc, uint32_t* rtxtAlloc, uint16_t* walk)
{
	uint32_t* rtxt = rtxtAlloc + BT_RTXT_PAD_WORDS;
	memset(rtxtAlloc, 0, (size_t)bt_rtxt_words(ix.len) * 4u);
	/* pass 1: rows in text order from the end; SA and the text.  Bowtie sorts the suffix "$" LAST (a suffix that is a prefix
	 * of another is the greater one): row len is the suffix at offset len */
	uint32_t row = ix.len;
	for (uint32_t p = ix.len; ; p--) {
		loc[row].x = p; loc[row].y = loc[row].z = loc[row].w = 0;
		if (p == 0) break;                                  /* row == zOff */
		uint32_t lf[4], L;
		bt_rank4(ix, row, lf, &L);
		const uint32_t y = ix.len - p;                      /* T[p-1] = base len-1-(p-1) of the reversed text */
		rtxt[y >> 4] |= L << (2u * (y & 15u));
		row = lf[L];
	}
	/* pass 2: the walk lengths by text offset -- 0 where the row is sampled (or is the '$' row), else one more than the
	 * offset before it.  Needs the row of every offset: from loc's SA column, inverted on the fly */
	{
		/* rowOf[p] is only needed here: reuse walk[] in two steps -- first mark sampled offsets */
		for (uint32_t r = 0; r <= ix.len; r++) {
			const uint32_t p = loc[r].x;
			walk[p] = ((r & ix.offMask) == r || r == ix.zOff) ? 0u : 1u;
		}
		uint32_t run = 0;
		for (uint32_t p = 0; p <= ix.len; p++) {
			if (walk[p] == 0u) run = 0; else run = run < 0xffffu ? run + 1u : run;
			walk[p] = (uint16_t)run;
		}
	}
	/* pass 3: every row's 48 characters of left context, from the reversed text */
	for (uint32_t r = 0; r <= ix.len; r++) {
		const uint32_t y0 = ix.len - loc[r].x, w = y0 >> 4, s = 2u * (y0 & 15u);
		uint32_t c[3];
		for (int k = 0; k < 3; k++) {
			const uint64_t two = (uint64_t)rtxt[w + k] | ((uint64_t)rtxt[w + k + 1] << 32);
			c[k] = (uint32_t)(two >> s);
		}
		loc[r].y = c[0]; loc[r].z = c[1]; loc[r].w = c[2];
	}
}

/* The image's pass 1 a CHAIN at a time (round 6; what the GPU loader runs, bt_kernels.hip: bt_loc_chain_kernel).  LF of the row
 * of text offset p is the row of offset p - 1, so from a sampled row S (SA[S] = s, the index's sample) the walk
 * S -> LF(S) -> ... visits the rows of offsets s, s - 1, s - 2, ... and ends where the next sampled row (or the '$' row,
 * offset 0) begins: every row lies on exactly one such chain -- the rows above the last sampled offset on one more, headed by
 * row `len` (the suffix at offset len) -- and the image costs one rank per ROW instead of one walk per row (2^offRate / 2
 * ranks: 15.5 at offRate 5, round 5's bt_loc_sa_kernel).  A row's walk length -- the reference's LF steps from it to a
 * sampled row, kept by text offset -- is its distance to the END of its chain: known once the chain has been gone through,
 * and written then (a contiguous stretch of walk[]).  rtxt must be zeroed beforehand. */
BT_HD uint64_t bt_loc_chain_count(const BtIndexDev& ix)
{
	return ((uint64_t)ix.len >> ix.offRate) + 1u + (((ix.len & ix.offMask) == ix.len) ? 0u : 1u);
}
BT_HD void bt_loc_chain(const BtIndexDev& ix, uint64_t i, BtU4* loc, uint32_t* rtxt, uint16_t* walk)
{
	const uint64_t nSampled = ((uint64_t)ix.len >> ix.offRate) + 1u;
	const bool head = i < nSampled;                                   /* the chain's first row is a sampled one */
	uint32_t row = head ? (uint32_t)(i << ix.offRate) : ix.len;
	const uint32_t s = !head ? ix.len : (row == ix.zOff ? 0u : BT_GP(const uint32_t, ix.offs)[row >> ix.offRate]);
	uint32_t k = 0;
	for (;;) {
		BT_GP(uint32_t, loc)[(uint64_t)row * 4u] = s - k;
		if (row == ix.zOff) break;                                   /* offset 0: nothing to the left, no LF; the chain's last row */
		uint32_t lf[4], L;
		bt_rank4(ix, row, lf, &L);
		const uint32_t y = ix.len - (s - k);                         /* T[sa-1] is base len-1-(sa-1) of the reversed text */
#if defined(__HIP_DEVICE_COMPILE__)
		atomicOr(rtxt + (y >> 4), L << (2u * (y & 15u)));
#else
		rtxt[y >> 4] |= L << (2u * (y & 15u));
#endif
		row = lf[L];
		k++;
		if ((row & ix.offMask) == row) break;                        /* the next chain's first row: its own lane's */
	}
	/* k = the distance from the chain's first row to its end (the next sampled row, or the '$' row): a row's walk length
	 * is its distance to that end -- 0 for a sampled first row */
	if (head) BT_GP(uint16_t, walk)[s] = 0;
	for (uint32_t j = head ? 1u : 0u; j < k; j++) { const uint32_t d = k - j; BT_GP(uint16_t, walk)[s - j] = (uint16_t)(d < 0xffffu ? d : 0xffffu); }
	if (row == ix.zOff) BT_GP(uint16_t, walk)[0] = 0;
}

#endif /* BT_WIDE: the side-layout block build and the locus image are the narrow build's */

/* ftabHi / ftabLo (ebwt.h:985-1034) */
BT_HD bt_row bt_ftab_hi(const BtIndexDev& ix, uint32_t i)
{
	bt_row v = BT_GP(const bt_row, ix.ftab)[i];
	if (v <= BT_ROWLIM(ix)) return v;
	return BT_GP(const bt_row, ix.eftab)[(v ^ BT_OFF_MASK) * 2u + 1u];
}
BT_HD bt_row bt_ftab_lo(const BtIndexDev& ix, uint32_t i)
{
	bt_row v = BT_GP(const bt_row, ix.ftab)[i];
	if (v <= BT_ROWLIM(ix)) return v;
	return BT_GP(const bt_row, ix.eftab)[(v ^ BT_OFF_MASK) * 2u];
}

#if !BT_WIDE
/* One entry of the jump table (BtIndexDev::jump): the ftab range of x's first ftabChars characters, then the reference's own
 * steps for the rest -- mapLF of both rows while the range has two or more (ebwt.h:2334-2380), mapLF1 while it has one
 * (:2494-2512) -- as GreedyDFSRangeSource::backtrack goes through positions it may not revisit (ebwt_search_backtrack.h:544-566). */
BT_HD void bt_jump_entry(const BtIndexDev& ix, uint32_t x, uint32_t K, uint32_t* topOut, uint32_t* botOut, uint32_t* metaOut)
{
	const uint32_t f = x & ((1u << (2u * ix.ftabChars)) - 1u);
	uint32_t top = bt_ftab_hi(ix, f), bot = bt_ftab_lo(ix, f + 1u);
	uint32_t nOps = 0, nMulti = 0, nSame = 0;
	for (uint32_t d = ix.ftabChars; d < K; d++) {
		if (bot <= top) break;
		const uint32_t c = (x >> (2u * d)) & 3u;
		uint32_t lf[4], L;
		nOps++;
		if (bot - top >= 2u) {
			nMulti++;
			if (top / 448u == bot / 448u) nSame++;
			bt_rank4(ix, top, lf, &L);
			const uint32_t t2 = lf[c];
			bt_rank4(ix, bot, lf, &L);
			top = t2; bot = lf[c];
		} else {
			bt_rank4(ix, top, lf, &L);
			if (L != c || top == ix.zOff) { top = 0; bot = 0; }
			else { top = lf[c]; bot = top + 1u; }
		}
	}
	if (bot <= top) { top = 0; bot = 0; }
	*topOut = top; *botOut = bot; *metaOut = nOps | (nMulti << 3) | (nSame << 6);
}
#endif

/* joinedToTextOff (ebwt.h:2569-2629): joined offset -> (tidx,toff); false if [off,off+qlen)
 * straddles a fragment boundary. */
BT_HD bool bt_joined_to_text(const BtIndexDev& ix, uint32_t qlen, bt_row off,
                             uint32_t* tidx, uint32_t* toff, uint32_t* probes)
{
	uint32_t top = 0, bot = ix.nFrag;
	for (;;) {
		uint32_t elt = top + ((bot - top) >> 1);
		bt_row lower = BT_GP(const bt_row, ix.rstarts)[elt * 3u];
		bt_row upper = (elt == ix.nFrag - 1u) ? ix.len : BT_GP(const bt_row, ix.rstarts)[(elt + 1u) * 3u];
		(*probes)++;
		if (lower <= off) {
			if (upper > off) {
				if (off + qlen > upper) return false;
				bt_row fraglen = upper - lower;
				bt_row fragoff = off - lower;
				if (!ix.fw) { fragoff = fraglen - fragoff - 1u; fragoff -= (qlen - 1u); }
				*tidx = (uint32_t)BT_GP(const bt_row, ix.rstarts)[elt * 3u + 1u];
				*toff = (uint32_t)(fragoff + BT_GP(const bt_row, ix.rstarts)[elt * 3u + 2u]);
				return true;
			}
			top = elt;
		} else {
			bot = elt;
		}
	}
}

#endif /* BT_RANK_H_ */
