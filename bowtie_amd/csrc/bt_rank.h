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

#define BT_OFF_MASK 0xffffffffu
#define BT_SIDE_SYMS 224u

/* device-visible image of one index (fw or mirror); all pointers are device pointers on the
 * GPU build, host pointers in the host unit-test build. */
struct BtIndexDev {
	const uint8_t*  ebwt;      /* numSidePairs * 128 bytes, reference byte layout            */
	const uint32_t* ftab;
	const uint32_t* eftab;
	const uint32_t* offs;
	const uint32_t* rstarts;   /* 3 * nFrag                                                  */
	const uint32_t* plen;
	uint32_t len, zOff, zSide, zSym, ftabChars, offRate, offMask, nFrag, fw, nPat;
	uint32_t fchr[5];
	uint32_t wide;             /* the index is a 64-bit (.ebwtl) build.  Its rows still fit 32 bits here, but the
	                              reference binary that serves it is compiled with 64-bit offsets, and two things
	                              a user can see follow the offset width: the row a hit is reported from is drawn
	                              with nextU<TIndexOffU>() (two draws, ebwt_search_backtrack.h:1538), and a
	                              best-first Branch is 160 bytes, so 1638 of them fit a pool chunk (pool.h:32)   */
};

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

/* Plain-load form: one lane fetches its own side (64 B) + the partner side's counters (8 B). */
BT_HD void bt_rank4(const BtIndexDev& ix, uint32_t row, uint32_t lf[4], uint32_t* L)
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

/* ftabHi / ftabLo (ebwt.h:985-1034) */
BT_HD uint32_t bt_ftab_hi(const BtIndexDev& ix, uint32_t i)
{
	uint32_t v = BT_GP(const uint32_t, ix.ftab)[i];
	if (v <= ix.len) return v;
	return BT_GP(const uint32_t, ix.eftab)[(v ^ BT_OFF_MASK) * 2u + 1u];
}
BT_HD uint32_t bt_ftab_lo(const BtIndexDev& ix, uint32_t i)
{
	uint32_t v = BT_GP(const uint32_t, ix.ftab)[i];
	if (v <= ix.len) return v;
	return BT_GP(const uint32_t, ix.eftab)[(v ^ BT_OFF_MASK) * 2u];
}

/* joinedToTextOff (ebwt.h:2569-2629): joined offset -> (tidx,toff); false if [off,off+qlen)
 * straddles a fragment boundary. */
BT_HD bool bt_joined_to_text(const BtIndexDev& ix, uint32_t qlen, uint32_t off,
                             uint32_t* tidx, uint32_t* toff, uint32_t* probes)
{
	uint32_t top = 0, bot = ix.nFrag;
	for (;;) {
		uint32_t elt = top + ((bot - top) >> 1);
		uint32_t lower = BT_GP(const uint32_t, ix.rstarts)[elt * 3u];
		uint32_t upper = (elt == ix.nFrag - 1u) ? ix.len : BT_GP(const uint32_t, ix.rstarts)[(elt + 1u) * 3u];
		(*probes)++;
		if (lower <= off) {
			if (upper > off) {
				if (off + qlen > upper) return false;
				uint32_t fraglen = upper - lower;
				uint32_t fragoff = off - lower;
				if (!ix.fw) { fragoff = fraglen - fragoff - 1u; fragoff -= (qlen - 1u); }
				*tidx = BT_GP(const uint32_t, ix.rstarts)[elt * 3u + 1u];
				*toff = fragoff + BT_GP(const uint32_t, ix.rstarts)[elt * 3u + 2u];
				return true;
			}
			top = elt;
		} else {
			bot = elt;
		}
	}
}

#endif /* BT_RANK_H_ */
