/*
 * bt_host.h -- host-side pieces behind the C ABI that need no GPU: the .ebwt loader and the
 * policy -> phase-program compiler.  Pure C++ (shared by libbowtie_amd.so and the unit tests).
 */
#ifndef BT_HOST_H_
#define BT_HOST_H_

#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/bowtie_amd.h"
#include "bt_core.h"
#include "bt_best.h"

/* One index (text or mirror) parsed into host memory.  Field order and geometry follow
 * Ebwt::readIntoMemory (ebwt.h:2926-3421) and EbwtParams::init (ebwt.h:138-184). */
struct BtIndexHost {
	bt_row   len = 0, zOff = 0;
	uint32_t nPat = 0, nFrag = 0;
	int32_t  lineRate = 0, linesPerSide = 0, offRate = 0, ftabChars = 0, flags = 0;
	bt_row   fchr[5] = {0, 0, 0, 0, 0};
	bool     fw = true;
	bool     wide = false;      /* came from a 64-bit (.ebwtl/.bt2l) build: see BtIndexDev::wide         */
	bool     bt2 = false;       /* came in bowtie2-build's side layout                                  */
	bool     swapped = false;   /* was written on a machine of the other byte order                      */
	std::vector<uint8_t>  ebwt;
	std::vector<bt_row> plen, rstarts, ftab, eftab, offs;
	std::vector<std::string> refnames;
	uint64_t ebwtTotLen() const { return ebwt.size(); }
#if BT_WIDE
	/* The wide build (64-bit BWT rows; bt_rank.h, "the row type") keeps no side layout: its loader derives the rank blocks
	 * and their segment table straight from the file's BWT, whatever the variant.
	 * rowBias (tests; BT_WIDE_ROW_BIAS, a multiple of 2^(segShift+6) and of 2^(offRate+1)): every BWT row of the image is numbered
	 * rowBias higher than in the files -- fchr, zOff, the ftab / eftab entries and the segment table carry it, and whoever
	 * binds the arrays to a BtIndexDev shifts the pointers of the row-indexed ones back by it (bt_host_index_bias) -- so that
	 * a genome of a few Mbp exercises row arithmetic above 2^32 (with BT_WIDE_SEG_SHIFT, blocks per segment log2, also the
	 * segments).  Text offsets are not biased.  Results are those of the unbiased index. */
	std::vector<uint8_t>  blk;
	std::vector<uint64_t> segBase;
	uint32_t segShift = BT_SEG_SHIFT_DEFAULT;
	uint64_t rowBias = 0;
#endif
};
#if BT_WIDE
/* after the pointers of a BtIndexDev were set to the start of the arrays: account for h.rowBias */
inline void bt_host_index_bias(const BtIndexHost& h, BtIndexDev* d)
{
	const uint64_t bb = h.rowBias / BT_BLK_ROWS;
	d->blk = (const uint8_t*)((uintptr_t)d->blk - (uintptr_t)(bb * BT_BLK_BYTES));
	d->segBase = (const uint64_t*)((uintptr_t)d->segBase - (uintptr_t)((bb >> h.segShift) * 32u));
	d->offs = (const bt_row*)((uintptr_t)d->offs - (uintptr_t)((h.rowBias >> h.offRate) * sizeof(bt_row)));
}
#endif

/* Returns BT_OK or BT_ERR_*.  offrate_override (>= index offRate) subsamples offs[] the way
 * -o does (ebwt.h:2988-3001, 3301-3327); -1 keeps the index's own rate. */
int bt_host_index_load(const std::string& base, bool fw, int offrate_override, BtIndexHost* out, int variant = -1);

/* The on-disk variant an index base names: 0 .bt2, 1 .ebwt, 2 .bt2l, 3 .ebwtl (-1: none found), in the
 * order the reference looks for them; and its file extension. */
int bt_host_index_variant(const std::string& base);
const char* bt_host_index_ext(int variant);
/* the text length the header of <base>.1.<ext> states (nothing else is read) */
int bt_host_index_header_len(const std::string& base, uint64_t* len64);

/* Fill the device-visible descriptor from host-side geometry (pointers are left to the caller). */
void bt_host_index_describe(const BtIndexHost& h, BtIndexDev* d);

/* Ebwt::restore (ebwt.h:2793-2824): invert the BWT of a loaded index into the joined reference
 * text (codes 0..3, `len` bytes).  Host-side utility (bowtie-inspect's job), not on the hot path. */
void bt_host_restore_text(const BtIndexHost& h, uint8_t* out);

/* Compile a bt_policy into the linear list of searcher invocations the reference's phase scripts
 * perform (search_exact.c, search_1mm_phase{1,2}.c, search_23mm_phase{1,2,3}.c,
 * search_seeded_phase{1..4}.c) with every policy-uniform condition (--nofw/--norc, seedMms)
 * resolved.  Returns BT_OK or BT_ERR_ARG. */
int bt_host_compile_program(const bt_policy& pol, BtProgram* prog);

/* The same for the stateful best-first workers (pol.best): the driver tree of
 * Unpaired{Exact,1mm,23mm,Seed}AlignerFactory::create() as a BfProgram (bt_best.h). */
int bt_host_compile_best(const bt_policy& pol, BfProgram* prog);
/* ... and of Paired*AlignerFactory::create() with --best (PairedBWAlignerV2): both mates' drivers in
 * one cost-aware driver, doubled sink limits, the RefAligner parameters. */
int bt_host_compile_best_paired(const bt_policy& pol, BfProgram* prog);

/* The 2-bit reference of <base>.3.ebwt / .4.ebwt (BitPairReference, reference.h:35-120) unpacked into
 * the position space BtRefDev describes.  Returns BT_OK or BT_ERR_IO / BT_ERR_FORMAT. */
struct BtRefHost {
	std::vector<uint32_t> bits, nmask, approxLen;
	std::vector<uint64_t> start;
};
int bt_host_ref_load(const std::string& base, const BtIndexHost& idx, BtRefHost* out, int variant = -1);

#endif
