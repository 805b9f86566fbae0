/*
 * bt_io.h -- host I/O either side of the search path: read files in,
 * SAM / default-format hit text out.  Plain host C++ (no HIP); the C entry points are declared in
 * include/bowtie_amd.h.  Behaviour follows the reference's pattern sources and hit sinks; the
 * citations are next to each function in bt_io.cpp.
 */
#ifndef BT_IO_H_
#define BT_IO_H_

#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/bowtie_amd.h"

/* a column of a batch: a view into the batch's one allocation (vector-like as far as the parsers need) */
template <typename T> struct BtCol {
	T* p = nullptr; size_t n = 0;
	T& operator[](size_t i) { return p[i]; }
	const T& operator[](size_t i) const { return p[i]; }
	T* data() { return p; }
	const T* data() const { return p; }
	size_t size() const { return n; }
	void resize(size_t m) { n = m; }          /* never beyond what BtHostBatch::reset sized it for */
};

/* Where batches get their memory (sequence and quality rows, lengths, seeds: what is uploaded).  Default: the C
 * library's.  The binary can hand in the library's pinned-memory allocator (bt_host_alloc / bt_host_free), which makes the
 * uploads of bt_align_stream_submit asynchronous.  Set before the first batch is made. */
void bt_io_set_allocator(void* (*alloc)(size_t bytes), void (*dealloc)(void* p));

/* one batch of parsed reads, laid out as bt_read_batch wants it (rows 16-byte aligned) */
struct BtHostBatch {
	uint32_t n = 0, stride = 16;
	uint64_t first_rdid = 0;              /* rdid[0]                                             */
	uint64_t end_rdid = 0;                /* read id the stream stands at after this batch: records taken from the input,
	                                         the ones that did not parse (and are not in the batch) included */
	uint8_t* seq = nullptr;               /* [cap][stride] codes 0..4, rows padded with 4        */
	uint8_t* qual = nullptr;              /* [cap][stride] Phred+33, rows padded with '!'        */
	BtCol<uint16_t> len;
	BtCol<uint32_t> seed;
	std::vector<uint64_t> rdid;           /* TReadId of each read (names default to it)          */
	std::vector<uint64_t> name_off;       /* n + 1 offsets into names                            */
	std::string names;
	std::vector<uint64_t> raw_off;        /* bt_read_opts.reserved bit 1: n + 1 offsets into raw   */
	std::string raw;                      /* each read's record as it stood in the input (Read::readOrigBuf) */
	uint32_t n_paired = 0;                /* BT_FMT_TABBED: reads whose record had a second end    */
	std::vector<uint8_t> paired;          /* BT_FMT_TABBED: per read, 1 = its record had a second end (empty for other formats) */
	size_t cap_bytes = 0;                 /* bytes of each of seq / qual inside `block` */
	uint32_t cap_reads = 0;               /* reads len / seed have room for                      */
	void* block = nullptr;                /* the one allocation: seq | qual | len | seed         */
	void (*block_free)(void*) = nullptr;  /* what releases it (the allocator it came from)       */

	BtHostBatch() {}
	~BtHostBatch();
	BtHostBatch(const BtHostBatch&) = delete;
	BtHostBatch& operator=(const BtHostBatch&) = delete;
	void reset(uint32_t n_reads, uint32_t stride_bytes);
	bt_read_batch view() const;
};

/* a stream of reads over one or more files (or the -c sequences) */
struct BtReadStream;
BtReadStream* bt_io_open(const char* spec, const bt_read_opts& opts, std::string* err);
/* next <= max_reads reads; returns BT_OK (batch.n == 0 at the end) or an error code with *err set */
int bt_io_next(BtReadStream* s, uint32_t max_reads, int threads, BtHostBatch* batch, std::string* err);
void bt_io_close(BtReadStream* s);
/* Two mate batches read side by side: keep the pairs both of whose mates parsed.  A record that does not parse drops
 * out of its own batch only; the reference parses the two mates of a read id together and skips the pair when either
 * fails (pat.cpp:96-127), so the batches are intersected on the read id.  Returns false if the two streams stand at
 * different read ids afterwards (one file has fewer records). */
bool bt_io_intersect_pairs(BtHostBatch* a, BtHostBatch* b);
/* A --12 file may hold pairs and unpaired reads (TabbedPatternSource, pat.cpp:977-1127: three fields or five).  a / b =
 * the batches of the two mate streams over that file (same records; an unpaired record's second end is empty).  Moves
 * the unpaired reads to `unp`, leaves the pairs in a / b, and says in `order` how the input interleaved them
 * (1 = the next pair, 0 = the next unpaired read).  false: the batch is inconsistent with itself (its column of pair flags
 * does not have one flag per record) and was left alone -- an internal error the caller reports as BT_ERR_READS. */
bool bt_io_split_tabbed(BtHostBatch* a, BtHostBatch* b, BtHostBatch* unp, std::vector<uint8_t>* order);

struct BtRefNames {
	std::vector<std::string> names;
	std::vector<uint32_t> lens;
};

/* appends the text for reads [lo, hi) of the batch to `out` and adds to the tally */
void bt_io_format(const bt_read_batch& rb, const char* names, const uint64_t* name_off, const bt_hit_batch& hb,
                  const BtRefNames& refs, const bt_out_opts& o, uint32_t lo, uint32_t hi, std::string* out,
                  bt_out_tally* tally);
/* the same for pairs (hit slots per pair hold adjacent mate alignments, upstream mate first) */
void bt_io_format_pairs(const bt_read_batch& r1, const char* names1, const uint64_t* off1,
                        const bt_read_batch& r2, const char* names2, const uint64_t* off2, const bt_hit_batch& hb,
                        const BtRefNames& refs, const bt_out_opts& o, uint32_t lo, uint32_t hi, std::string* out,
                        bt_out_tally* tally);
void bt_io_sam_header(const BtRefNames& refs, const bt_out_opts& o, const char* cmdline, const char* rgline,
                      std::string* out);
void bt_io_summary(const bt_out_tally& t, std::string* out);

#endif /* BT_IO_H_ */
