/*
 * bowtie_amd.h -- C ABI of the MI355X-native FM-index search hot path.
 *
 * This is the drop-in boundary for the one data-parallel path of BenLangmead/bowtie v1.3.1 that
 * this project accelerates: everything a reference worker thread does for a read between
 * GET_READ and FINISH_READ (ebwt_search.cpp:923-961).  The phase-program engine (default, non --best,
 * unpaired search modes):
 *
 *     exactSearchWorker                    ebwt_search.cpp:1130   (-v 0)
 *     mismatchSearchWorkerFull             ebwt_search.cpp:1606   (-v 1)
 *     twoOrThreeMismatchSearchWorkerFull   ebwt_search.cpp:2056   (-v 2)
 *     seededQualSearchWorkerFull           ebwt_search.cpp:2378   (-n 0..3 -l -e)
 *
 * i.e. GreedyDFSRangeSource::backtrack (ebwt_search_backtrack.h:237-1091), the Ebwt rank/LF
 * primitives (ebwt.h:1418-1523, 1696-2560), Ebwt::reportChaseOne/joinedToTextOff
 * (ebwt.h:2569-2755), the phase scripts search_*.c and the per-read stop/continue policy of
 * NGoodHitSinkPerThread / AllHitSinkPerThread (hit.h:969-985, 1201-1209); and the best-first engine
 * (`--best`, `--strata`, `-M`, `-v 3`, and every paired-end run): the *Stateful workers
 * (ebwt_search.cpp:1223, 1509, 1955, 2609) with UnpairedAlignerV2, PairedBWAlignerV1 (pairs without --best,
 * aligner.h:606-1480) and PairedBWAlignerV2 (pairs with --best, aligner.h:1483-2051).
 *
 * The reference has no FFI for this path (it is one C++ process); the seams this ABI replaces
 * are cited per entry point.  Plain pointers and sizes only; no exceptions cross the boundary
 * (the reference signals errors with `throw 1`, ebwt_search.cpp:3449-3461; here: int codes).
 *
 * All arithmetic on the path is integer (u8/u32/u64 + popcount).  Results are bit-identical to
 * the reference's for the same reads, index and policy.
 */
#ifndef BOWTIE_AMD_H_
#define BOWTIE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------ */
#define BT_OK              0
#define BT_ERR_IO          1   /* index file missing / short read                             */
#define BT_ERR_FORMAT      2   /* not a bowtie index family member this loader reads (.ebwt,
                                  .ebwtl / .bt2l with < 2^32-1 rows, .bt2 side layout), or damaged */
#define BT_ERR_ARG         3   /* bad policy / batch                                          */
#define BT_ERR_DEVICE      4   /* HIP runtime error (no GPU, OOM, launch failure)             */
#define BT_ERR_READ_SHORT  5   /* read shorter than the mode allows (reference: throw 1,
                                  search_1mm_phase1.c:12-15, search_23mm_phase1.c:13-20)      */
#define BT_ERR_OVERFLOW    6   /* a read exceeded a per-read scratch capacity (status bit
                                  BT_ST_OVERFLOW / BT_ST_MMPOOL says which reads); the other
                                  reads of the batch are valid                                */
#define BT_ERR_READS       7   /* malformed read input (the reference prints a message and
                                  exits 1; the message is in bt_reads_error())                */
#define BT_ERR_ROWS64      8   /* the index has 2^32-1 BWT rows or more: it needs the build with 64-bit rows,
                                  libbowtie_amd_l.so / bowtie-amd-l (the reference's bowtie-align-l, btypes.h:4-28) */
#define BT_ERR_UNSUPPORTED 9   /* not in this build: the 64-bit-row build has no 32-bit probes and no gather benchmark
                                  (bt_probe_rank64 is its probe); both engines and pairs are in both builds       */

/* ---- policy: exactly the knobs the reference workers read -------------------------------- */
#define BT_MODE_V 0            /* end-to-end, -v <mms>   (ebwt_search.cpp:3249-3268)          */
#define BT_MODE_N 1            /* seeded/quality-aware, -n <mms> -l -e (ebwt_search.cpp:3243) */

typedef struct bt_policy {
	int32_t  mode;         /* BT_MODE_V | BT_MODE_N                                           */
	int32_t  mms;          /* -v 0..2  or  -n 0..3 (seedMms)                                  */
	int32_t  seed_len;     /* -l (default 28)                     ebwt_search.cpp:162         */
	int32_t  qual_thresh;  /* -e (default 70)                     ebwt_search.cpp:163         */
	int32_t  max_bts;      /* --maxbts (default 125; half-and-half searchers only enforce it,
	                          ebwt_search_backtrack.h:428-434)                                */
	int32_t  nofw;         /* --nofw                                                          */
	int32_t  norc;         /* --norc                                                          */
	int32_t  maq_round;    /* 1 unless --nomaqround               qual.cpp:4                  */
	uint32_t khits;        /* -k (default 1)                      hit.h:969-985               */
	uint32_t mhits;        /* -m (0xffffffff = unlimited)                                     */
	int32_t  all_hits;     /* -a                                  hit.h:1201-1209             */
	int32_t  best;         /* 1 = the reference's stateful best-first workers (--best; implied by
	                          --strata, -M and -v 3: ebwt_search.cpp:775-776, 851-853, 877-882);
	                          max_bts then defaults to 800 (ebwt_search.cpp:186)                */
	int32_t  strata;       /* --strata                            hit.h:1070-1129             */
	int32_t  sample_max;   /* -M: reads over mhits keep their first mhits hits so that one can
	                          be sampled (hit.cpp:16-68, sam.cpp:263-311)                      */
	/* paired-end (bt_align_pairs; PairedBWAlignerV2, aligner.h:1483-2051) */
	int32_t  min_ins;      /* -I (0)     already reduced by the mates' 5'/3' trimming          */
	int32_t  max_ins;      /* -X (250)   (aligner.h:1921-1935)                                 */
	int32_t  mate1_fw;     /* --fr: 1,0  --ff: 1,1  --rf: 0,1        ebwt_search.cpp:902-906   */
	int32_t  mate2_fw;
	int32_t  pair_tries;   /* --pairtries (100): anchors tried per pair (aligner.h:1855)       */
	int32_t  allow_contain;/* --allow-contain                                                  */
	int32_t  pe_v1;        /* pairs only: 1 = PairedBWAlignerV1 (aligner.h:606-1480), the reference's paired-end
	                          aligner when --best is NOT given (ebwt_search.cpp:232, 776); 0 = PairedBWAlignerV2
	                          (--best).  Either way the stateful engine runs (best is taken as set) and max_bts
	                          defaults to 800 (ebwt_search.cpp:186, 2644, 2670) */
	int32_t  reserved[1];
} bt_policy;

void bt_policy_default(bt_policy* p);   /* reference defaults: -n 2 -l 28 -e 70 -k 1          */
int  bt_has_pe_v1(void);                /* 1: this build of the library takes bt_policy.pe_v1 (every build since round 3) */

/* ---- reads in: what PatternSourcePerThread hands the worker (read.h:42-273) -------------- */
typedef struct bt_read_batch {
	uint32_t        n_reads;
	uint32_t        stride;   /* bytes per read row in seq[] and qual[]: >= max len, a multiple
	                             of 16; seq and qual themselves 16-byte aligned               */
	const uint8_t*  seq;      /* [n_reads][stride]  A=0 C=1 G=2 T=3 N=4 (patFw)               */
	const uint8_t*  qual;     /* [n_reads][stride]  Phred+33 ASCII (Read::qual)               */
	const uint16_t* len;      /* [n_reads]          1..1024                                   */
	const uint32_t* seed;     /* [n_reads]          Read::seed = genRandSeed (pat.cpp:21-57)  */
} bt_read_batch;

/* ---- hits out: the fields of Hit the search computes (hit.h:56-112, ebwt.h:1288-1405) ---- */
typedef struct bt_hit {
	uint32_t tidx;            /* Hit::h.first   reference sequence index                      */
	uint32_t toff;            /* Hit::h.second  0-based offset of leftmost base               */
	uint32_t oms;             /* Hit::oms       bot-top-1                                     */
	uint32_t mm_off;          /* first entry of this hit's mismatches in mm_pool              */
	uint16_t cost;            /* Hit::cost      ham | stratum<<14                             */
	uint16_t nmm;             /* # mismatches                                                 */
	uint8_t  stratum;         /* Hit::stratum                                                 */
	uint8_t  fw;              /* Hit::fw                                                      */
	uint8_t  pad[2];          /* pad[0] = Hit::mate: 0 unpaired, 1 / 2 mate of a paired alignment */
} bt_hit;                         /* 24 bytes */

/* mm_pool entry: bits 0-9 = offset from the read's 5' end (Hit::mms), bits 12-13 = reference
 * base at that offset in the orientation reported by the reference (Hit::refcs).            */
#define BT_MM_POS(e)  ((uint32_t)(e) & 0x3ffu)
#define BT_MM_REFC(e) (((uint32_t)(e) >> 12) & 3u)

/* per-read status bits */
#define BT_ST_SKIPPED   1u   /* -n mode: len<4 or too many Ns in seed (search_seeded_phase1.c:17-44) */
#define BT_ST_HITCAP    2u   /* more reportable hits than hit_cap slots (slots hold the first ones)  */
#define BT_ST_TOOSHORT  4u   /* -v 1: len<2, -v 2: len<4 (the reference aborts the whole run)        */
#define BT_ST_OVERFLOW  8u   /* backtrack-frame / range-stack / seedling capacity exceeded           */
#define BT_ST_MMPOOL    16u  /* mm_pool exhausted: hit stored with nmm = 0                           */

typedef struct bt_hit_batch {
	uint32_t  hit_cap;        /* hit slots per read                                           */
	bt_hit*   hits;           /* [n_reads][hit_cap]                                           */
	uint32_t* n_hits;         /* [n_reads] HitSinkPerThread::hitsForThisRead_ at finishRead   */
	uint8_t*  status;         /* [n_reads] BT_ST_*                                            */
	uint16_t* mm_pool;        /* [mm_pool_cap]                                                */
	uint32_t  mm_pool_cap;
	uint32_t  mm_pool_used;   /* out                                                          */
} bt_hit_batch;

/* operation counters of one batch (what the reference counts under -DEBWT_STATS,
 * ebwt.h:2343-2345, 2424-2426, 2462-2464, 2498-2500): feed the algorithmic-bytes model.     */
typedef struct bt_op_counts {
	uint64_t lfex;            /* two-locus all-char rank steps (mapLFEx)                      */
	uint64_t lf2;             /* two-locus single-char steps (mapLF(ltop,c)+mapLF(lbot,c))    */
	uint64_t lf1;             /* single-locus steps (mapLF1)                                  */
	uint64_t chase;           /* SA-walk steps (mapLF(l) in reportChaseOne)                   */
	uint64_t ftab;            /* ftab lookups (pairs)                                         */
	uint64_t offs;            /* offs[] lookups                                               */
	uint64_t rstarts;         /* rstarts probes                                               */
	uint64_t frames;          /* backtrack frames entered                                     */
	uint64_t lane_iters;      /* sum over lanes of lock-step iterations (GPU only)            */
	uint64_t same_pair;       /* two-locus steps whose rows share one 128-byte side pair      */
	uint64_t rescans;         /* frame re-scans for the next-lowest-quality target set        */
	uint64_t cand_scans;      /* frame scans for the deepest remaining target                 */
	uint64_t wave_rounds;     /* lock-step rounds summed over wavefronts (GPU only);
	                             lane_iters / wave_rounds = mean active lanes per round        */
	uint64_t fetches;         /* rounds a lane spent on a non-rank request (read window, backtrack
	                             target, frame record, record scan, ftab, SA sample)           */
	/* Locus mode (round 5; csrc/bt_rank.h "the locus image"): once a range is one BWT row the steps above are decided by
	 * comparing the read with the text, and a reported row's offset comes from a dense suffix array.  The reference's
	 * steps that were decided that way are COUNTED IN lfex / lf1 / chase above all the same (and a one-row mapLFEx in
	 * same_pair: its two rows are neighbours; one in 448 such pairs straddles two side pairs, which row space would
	 * have known and the text does not); these say how many of them, and what was fetched instead. */
	uint64_t loc_lfex;        /* mapLFEx steps on a one-row range decided by the text           */
	uint64_t loc_lf1;         /* mapLF1 steps decided by the text                               */
	uint64_t loc_chase;       /* SA-walk steps not walked (from the table of walk lengths)      */
	uint64_t loc_records;     /* 16-byte locus records fetched (one per range that became one row, one per reported row) */
	uint64_t loc_windows;     /* text windows fetched (32 bytes: the read's remaining length)   */
} bt_op_counts;

/* index geometry, for callers that need it (EbwtParams, ebwt.h:116-321) */
typedef struct bt_index_info {
	uint32_t len, n_pat, n_frag, ftab_chars, off_rate, z_off;
	uint64_t ebwt_bytes, offs_bytes;
	int32_t  has_mirror;
	int32_t  variant;       /* BT_INDEX_* of the files the image was loaded from, | BT_INDEX_SWAPPED */
} bt_index_info;

/* The .ebwt family (ebwt.h:138-183, 2926-2973; btypes.h:4-28).  bt_index_load looks for them in the
 * reference's order -- <base>.1.bt2, .1.ebwt, then the 64-bit builds .1.bt2l, .1.ebwtl (adjustEbwtBase
 * ebwt.cpp:36-48; the `bowtie` wrapper's choice of the -l binary, bowtie:52-81) -- in either byte
 * order, and converts what it finds to one in-memory layout.  64-bit builds load when the index has
 * fewer than 2^32-1 rows (BT_ERR_ROWS64 otherwise: libbowtie_amd_l.so, the same sources compiled with
 * 64-bit rows, holds those -- bt_rows64() tells the two libraries apart), and keep the two behaviours
 * of the 64-bit binary that a user can see (see BtIndexDev::wide in csrc/bt_rank.h). */
#define BT_INDEX_BT2      0
#define BT_INDEX_EBWT     1
#define BT_INDEX_BT2L     2
#define BT_INDEX_EBWTL    3
#define BT_INDEX_SWAPPED  16

typedef struct bt_index bt_index;   /* device-resident fw (+ mirror) index image              */
typedef struct bt_ctx   bt_ctx;     /* per-GPU stream, scratch, queues                        */

/* Replaces: Ebwt ctor + Ebwt::loadIntoMemory for <base>.{1,2}.ebwt and, if need_mirror,
 * <base>.rev.{1,2}.ebwt (ebwt.h:402-448, 2835-3445; ebwt_search.cpp:3120-3155); then uploads
 * the arrays to the HBM of `device`.  offrate_override = -1 keeps the index's offRate (-o). */
int  bt_index_load(const char* ebwt_base, int need_mirror, int offrate_override, int device,
                   bt_index** out);
void bt_index_info_get(const bt_index* idx, bt_index_info* info);
/* 1 in the library built with 64-bit BWT rows (libbowtie_amd_l.so), 0 in libbowtie_amd.so; the text length of an
 * index whatever its size (bt_index_info::len is 32 bits: 0xffffffff there if it does not fit) */
int      bt_rows64(void);
uint64_t bt_index_len64(const bt_index* idx);
const char* bt_index_refname(const bt_index* idx, uint32_t tidx);   /* Ebwt::refnames()      */
uint32_t    bt_index_reflen (const bt_index* idx, uint32_t tidx);   /* Ebwt::plen()          */
void bt_index_free(bt_index* idx);
/* Host-side utility (no GPU): joined reference text of <base>.1.ebwt as codes 0..3, `len` bytes
 * (Ebwt::restore, ebwt.h:2793-2824; what bowtie-inspect prints).  cap = bytes available in out. */
int  bt_index_restore_text(const char* ebwt_base, uint8_t* out, uint64_t cap);
/* Host-side utility (no GPU): out[0] = BT_INDEX_* (| BT_INDEX_SWAPPED) found for the base, out[1] = text
 * length, out[2..7] = FNV-1a-64 digests of the loaded image's arrays (ebwt, ftab, eftab, offs,
 * plen+rstarts, zOff/fchr/offRate/ftabChars).  Equal digests = the same image, whatever the files. */
int  bt_index_digest(const char* ebwt_base, int mirror, uint64_t out[8]);
/* Host-side utility (no GPU), a few bytes of the header read: 1 = the index has 2^32-1 rows or more and needs the
 * build with 64-bit rows, 0 = it does not, < 0 = BT_ERR_IO / BT_ERR_FORMAT.  Replaces: the file-name test by which
 * the reference's wrapper picks bowtie-align-l (bowtie:52-81) -- decided before any read is taken from the input. */
int  bt_index_needs_rows64(const char* ebwt_base);

/* Replaces: the per-thread set-up at the top of each worker (sink, params, 1..9
 * GreedyDFSRangeSource objects; ebwt_search.cpp:1155, 2082-2134, 2413-2539).  One ctx per GPU,
 * driven by one host thread.  `stream` = a hipStream_t to launch on (NULL = own stream).   */
int  bt_ctx_create(const bt_index* idx, const bt_policy* pol, void* stream, bt_ctx** out);
void bt_ctx_destroy(bt_ctx* ctx);

/* Replaces: the body of the worker loop for a batch of reads (FINISH_READ/GET_READ/#include
 * "search_*.c"; ebwt_search.cpp:1183-1199, 1660-1682, 2151-2176, 2557-2585).
 *   bt_align_batch         : host pointers in `in`/`out`; copies H2D, runs, copies D2H.
 *   bt_align_batch_device  : every pointer in `in`/`out` is a device pointer (reads already in
 *                            HBM, hits stay in HBM); asynchronous on the ctx stream.  `counts_dev`
 *                            must be NULL since 0.2.0 (BT_ERR_UNSUPPORTED otherwise): the op counters
 *                            of the batches since the last reset are read with bt_ctx_counts after
 *                            bt_ctx_sync.
 *                            Stream order is the caller's: a context's own stream (stream == NULL at
 *                            bt_ctx_create) is NON-BLOCKING -- it does not wait for the null stream -- so
 *                            whatever produced the arrays on another stream (a fill, a copy, a kernel)
 *                            must be complete, or the caller's stream must be the one the context was
 *                            created on. */
int  bt_align_batch(bt_ctx* ctx, const bt_read_batch* in, bt_hit_batch* out, bt_op_counts* counts);
int  bt_align_batch_device(bt_ctx* ctx, const bt_read_batch* in, bt_hit_batch* out,
                           bt_op_counts* counts_dev);
/* Paired-end (-1/-2 with --best).  Replaces: BitPairReference's constructor (reference.h:35-240) --
 * <base>.3.ebwt / .4.ebwt into HBM -- and, per batch of pairs, PairedBWAlignerV2::setQuery/advance
 * (aligner.h:1571-1701) with RefAligner::find (ref_aligner.h:63-101) for the second mate.  The ctx
 * must have been created with pol.best; the pair options are pol.min_ins .. pol.allow_contain.
 * in1/in2: mates 1 and 2, same n_reads.  out: hit_cap slots per *pair* (even): the alignments of a
 * pair are adjacent, upstream mate first, bt_hit.pad[0] = 1 or 2 says which mate a record is; n_hits
 * counts mate alignments (two per reported pair), as the reference's sink does. */
int  bt_index_load_reference(bt_index* idx);
int  bt_align_pairs(bt_ctx* ctx, const bt_read_batch* in1, const bt_read_batch* in2, bt_hit_batch* out,
                    bt_op_counts* counts);
int  bt_align_pairs_device(bt_ctx* ctx, const bt_read_batch* in1, const bt_read_batch* in2, bt_hit_batch* out,
                           bt_op_counts* counts_dev);
/* A stream of host batches through one context: bt_align_stream_submit uploads a batch and enqueues its search
 * without waiting (PCIe traffic on a copy stream of its own, a staging area in HBM per batch in flight);
 * bt_align_stream_collect hands back the oldest submitted batch if it is complete -- its bt_hit_batch filled,
 * `*tag` = the value given at submit -- and `*tag = NULL` otherwise (not complete yet, or nothing in flight): the
 * call does not wait unless flush != 0, which finishes whatever is still being searched (end of input).  At most
 * 62 batches may be in flight (BT_BATCH_RING - 2; 14 in rounds 2-5).  Reads that outgrow the search scratch come back flagged BT_ST_OVERFLOW (no second
 * pass on the stream); run them through bt_align_batch.  `in`, `out` and the arrays they point at stay the caller's
 * and must live until the batch is collected.  What the reference does with a FASTQ reader feeding its worker
 * threads.  (Measured on MI355X, round 6: a copy does not start while a launch's persistent kernel holds the device's
 * registers -- uploads and results move in the gaps between launches, some 0.05 s of a 0.77 s cadence at 12 M reads per
 * batch -- which is why a collect with flush == 0 never waits for them: poll.) */
/* Locus mode (csrc/bt_rank.h "the locus image"): the phase-program engine replaces, from the step at which a
 * range is one BWT row, the reference's row-by-row mapLF1 / mapLFEx steps (ebwt_search_backtrack.h:544-566, ebwt.h:2334-2380,
 * 2494-2512) by comparisons with the text, and the SA walk of a reported row (Ebwt::reportChaseOne, ebwt.h:2693-2755) by a
 * look-up in a dense suffix array -- same results, same op counts (bt_op_counts).  The image (18.25 bytes per base and
 * index) is derived on the device when the first such context is created, if the device has the room and BT_LOCUS=0 does not
 * forbid it.  bt_ctx_set_locus switches a context's launches between the two ways (A/B, diagnostics). */
int      bt_ctx_set_locus(bt_ctx* ctx, int on);
int      bt_ctx_get_locus(const bt_ctx* ctx);
uint64_t bt_index_locus_bytes(const bt_index* idx);          /* 0: the index has no locus image */
double   bt_index_locus_build_seconds(const bt_index* idx);
int      bt_index_locus_copy(const bt_index* idx, int mirror, void* loc, void* rtxt, void* walk);   /* tests: the image's arrays to host buffers */
int  bt_align_stream_submit(bt_ctx* ctx, const bt_read_batch* in, bt_hit_batch* out, void* tag);
/* *batches = how many batches of this shape (reads, stride, hit_cap, mm_pool_cap) the device has room for in flight: half of
 * its free memory now, in staging areas of that size, plus the context's idle ones.  For a caller that chooses how many
 * batches to let ride (the reference's equivalent is the number of worker threads: each holds one read at a time). */
int  bt_align_stream_room(bt_ctx* ctx, const bt_read_batch* in, const bt_hit_batch* out, uint32_t* batches);
int  bt_align_stream_collect(bt_ctx* ctx, void** tag, int flush);
/* With carry-over: one more launch of the context's grid with no new reads.  What is parked runs on for at least
 * `min_rounds` lock-step rounds (0: the library's default) or to its end, is parked again, and what it completes is known as
 * after any launch -- bt_align_stream_collect(ctx, &tag, 0) then hands out the batches that are done.  For the end of the
 * input: a few ticks let the oldest batches out one by one while the stragglers of the later ones still run, where a
 * flush holds every batch back until the last straggler of the last one is done (the reference's threads finish their
 * reads in whatever order they come, ebwt_search.cpp:1180-1230; its output queue does the re-ordering, hit.h:600-700).
 * Asynchronous like a submit; a context without carry-over, or with nothing parked, is left alone. */
int  bt_align_stream_tick(bt_ctx* ctx, uint32_t min_rounds);
/* Page-locked host memory for the arrays of a bt_read_batch handed to bt_align_stream_submit: from such memory the
 * upload is a DMA the call does not wait for (from ordinary memory the runtime stages it and the call returns when it
 * is done).  NULL when there is none to be had; ordinary memory works everywhere.  (A reader thread's parse buffers:
 * the reference's PatternSourcePerThread owns the equivalent, pat.h:163-201.) */
void* bt_host_alloc(size_t bytes);
void  bt_host_free(void* p);
int  bt_ctx_sync(bt_ctx* ctx);
/* Carry-over between the batches of a context (what the reference's worker threads get for free: a thread that
 * finishes its read takes the next one, whatever "batch" it came from -- ebwt_search.cpp:1180-1230's GET_READ loop).
 * launches = 0 (default): every bt_align_batch_device call runs its batch to the last read before the next one
 * starts.  launches = n (1..62; 1..14 in rounds 2-5): when a batch's reads have all been handed out, the searches still running are
 * parked and the following call on this context resumes them alongside its own reads -- a read may ride along for
 * up to n launches, after which the launch it is in finishes it -- so the minority of reads that backtrack for a
 * long time never leave the GPU idle.  The contract changes accordingly: the results of a batch are complete when
 * the stream work of the n-th bt_align_batch_device call after its own is, or after bt_ctx_sync (which finishes
 * whatever is parked); the batch's input and output arrays must stay valid until then.  Reads <= 112 bases,
 * unpaired, phase-program engine; other batches are simply run to completion as before.
 * bowtie-amd and bench.py (steps under 64 M reads) use it by default since round 3.
 * Measured gain: 8.44 M against 4.3 M reads/s at 16 M reads per batch (round 2). */
int  bt_ctx_set_carry(bt_ctx* ctx, int launches);
/* bt_align_batch_device sees the read lengths in HBM only; which build of the kernel a batch can use (reads kept
 * in LDS up to 104 / 112 bases, ebwt_search_backtrack.h:90-140's query accessors) then has to be settled on the
 * device.  A driver that knows its reads (a sequencer's fixed length) says so here: max_len = the longest read of
 * the batches to come, 0 = unknown again.  A longer read is not searched and is flagged BT_ST_OVERFLOW. */
int  bt_ctx_set_max_read_len(bt_ctx* ctx, uint32_t max_len);
/* timing of the launches since the previous bt_ctx_sync (call after the next one): total milliseconds from the first
 * launch's start to the last one's end and the number of batches; the i-th batch's own launches (the last 16 are
 * kept; i = -1: the closing flush of parked reads, 0 without carry-over) */
float bt_ctx_span_ms(bt_ctx* ctx, uint32_t* n_launches);
float bt_ctx_launch_ms(bt_ctx* ctx, int i);
uint32_t bt_ctx_last_carried(bt_ctx* ctx);   /* after bt_ctx_sync: reads the last two launches parked (diagnostics) */
/* The jump table (round 6): where the phase may not revisit a search's first 14 characters, the range behind them comes from
 * one look-up in a table derived at load (10 bytes per 14-mer: 2.7 GB per index; genomes of 4 Mbp and more; BT_JUMP_CHARS=0:
 * none) instead of ftab's 10 characters and four dependent LF steps.  The op counters still say what the reference's algorithm
 * does.  bt_ctx_jump_counts (after bt_ctx_counts): the look-ups among the counted searches and the LF steps behind them. */
void     bt_ctx_jump_counts(bt_ctx* ctx, uint64_t* lookups, uint64_t* steps);
uint64_t bt_index_jump_bytes(const bt_index* idx);
/* after bt_ctx_sync: mm_pool entries the last device-pointer batch used */
uint32_t bt_ctx_last_mm_used(bt_ctx* ctx);
/* reads of the last bt_align_batch that outgrew their search arenas and were re-run with worst-case
 * arenas (the reference sizes every backtrack frame for the whole read instead, ebwt_search_backtrack.h:107) */
uint32_t bt_ctx_last_retried(bt_ctx* ctx);
/* diagnostics (profiling build of the library only; zeros otherwise): wavefront cycles spent per
 * section of the automaton since the counters were last reset */
int  bt_ctx_prof_sections(bt_ctx* ctx, uint64_t* out, int n);
/* diagnostics: device buffer [n_reads] that receives, per read, the number of lock-step LF rounds it
 * took (NULL = off; stays set for later batches) */
void bt_ctx_set_iters_buffer(bt_ctx* ctx, uint32_t* dev_ptr);
/* read (and optionally reset) the ctx-owned op counters that bt_align_batch_device accumulates
 * into when counts_dev == NULL */
int  bt_ctx_counts(bt_ctx* ctx, bt_op_counts* out, int reset);
/* milliseconds the search kernel(s) of the last bt_align_batch[_device] call took, measured with
 * HIP events on the ctx stream (valid after bt_ctx_sync). */
float bt_ctx_last_kernel_ms(bt_ctx* ctx);
/* name of the kernel (template instance) the last batch ran, as a profiler lists it */
const char* bt_ctx_last_kernel_name(bt_ctx* ctx);

const char* bt_strerror(int code);
const char* bt_version(void);

/* ---- kernel-level probes (known-answer tests against the reference's Ebwt methods) -------- */
/* rows[n] -> lf[n][4] = mapLFEx (ebwt.h:2334), L[n] = rowL (ebwt.h:1696).  mirror bit 0: probe the .rev index;
 * bit 1: rank from the index files' 224-symbol side layout instead of the 32-byte rank blocks the search kernels
 * query (bt_rank.h) -- the two must agree.  Host pointers. */
int bt_probe_rank(bt_ctx* ctx, int mirror, const uint32_t* rows, uint32_t n, uint32_t* lf, uint8_t* L);
/* the same with rows as 64-bit numbers, from the rank blocks -- in either library; in libbowtie_amd_l.so (bt_rows64()) the
 * three 32-bit probes around it answer BT_ERR_UNSUPPORTED */
int bt_probe_rank64(bt_ctx* ctx, int mirror, const uint64_t* rows, uint32_t n, uint64_t* lf, uint8_t* L);
/* rows[n] -> joined-text offset via the SA walk of reportChaseOne (ebwt.h:2727-2746) and
 * (tidx,toff) via joinedToTextOff (ebwt.h:2569) for a query of length qlen; tidx=0xffffffff when
 * the hit straddles a fragment boundary. */
int bt_probe_chase(bt_ctx* ctx, int mirror, const uint32_t* rows, uint32_t n, uint32_t qlen,
                   uint32_t* joined_off, uint32_t* tidx, uint32_t* toff);

/* Measurement aid: the random-gather ceiling of this GPU on this index -- n_blocks x 256 lanes each
 * doing `iters` rank queries at pseudo-random rows (dependent != 0: each row derived from the previous answer, as in an
 * SA walk), timed with HIP events.  mirror bit 0: the .rev index; bit 1: gather 128-byte side pairs of the index files'
 * layout instead of the 32-byte rank blocks the search kernels gather.  *gbs_out counts 32 (128) bytes per query. */
int bt_bench_gather(bt_ctx* ctx, int mirror, uint32_t n_blocks, uint32_t iters, int dependent,
                    float* ms, double* gbs);

/* ---- host I/O either side of the path --------------------------------
 * Read files -> bt_read_batch, bt_hit_batch -> the reference's output text.  Host-only code in
 * the same library; no GPU is needed to call these. */
#define BT_FMT_FASTQ    0   /* -q (default)  FastqPatternSource        pat.cpp:797-975          */
#define BT_FMT_FASTA    1   /* -f            FastaPatternSource        pat.cpp:531-640          */
#define BT_FMT_RAW      2   /* -r            RawPatternSource          pat.cpp:1129-1213        */
#define BT_FMT_CMDLINE  3   /* -c            VectorPatternSource       pat.cpp:359-528          */
#define BT_FMT_FASTA_CONT 4 /* -F <len>,<freq>  FastaContinuousPatternSource pat.cpp:651-793     */
#define BT_FMT_TABBED   5   /* --12          TabbedPatternSource       pat.cpp:977-1127: one record per line,
                               "name\tseq\tquals" or "name\tseq1\tquals1\tseq2\tquals2"; with BT_READ_MATE2 the
                               second end is delivered, otherwise the first (bt_reads_paired_count tells which
                               records had two)                                                       */
#define BT_QUAL_PHRED33  0  /* charToPhred33, qual.h:89-127                                      */
#define BT_QUAL_PHRED64  1  /* --phred64-quals / --solexa1.3-quals                               */
#define BT_QUAL_SOLEXA64 2  /* --solexa-quals                                                    */
#define BT_QUAL_INT      3  /* --integer-quals: space-separated Phred integers (FASTQ only;
                               intToPhred33, qual.h:132-153; pat.cpp:918-936)                  */
#define BT_QUAL_INT_SOLEXA 4 /* --integer-quals --solexa-quals                                   */
#define BT_READ_CAREFUL  1u /* every record through the step-by-step parser (testing)            */
#define BT_READ_KEEP_RAW 2u /* keep each read's record text (Read::readOrigBuf) for --al/--un/--max */
#define BT_READ_MATE1    4u /* the file holds first mates (-1): names end in /1 -- appended unless there --  */
#define BT_READ_MATE2    8u /* ... second mates (-2), /2; the seed covers the name (read.h:141-165, pat.cpp:76-88) */
#define BT_READ_INTERLEAVED 16u /* --interleaved (FASTQ): records alternate first mate, second mate (pat.cpp:797-856);
                                   with BT_READ_MATE1 / BT_READ_MATE2 the even / odd records are delivered, a
                                   trailing record without its mate is dropped                          */

typedef struct bt_read_opts {
	int32_t  format;       /* BT_FMT_*                                                        */
	int32_t  trim5, trim3; /* -5 / -3                         pat.h TrimmingPatternSource      */
	int32_t  qual_enc;     /* BT_QUAL_*                                                       */
	uint32_t seed;         /* --seed: mixed into every read's seed (pat.cpp:21-57)            */
	uint32_t flags;        /* BT_READ_* bits                                                  */
	uint64_t skip;         /* -s: first reads to skip         pat.cpp:113-115                 */
	uint64_t upto;         /* -u: reads to process after the skipped ones (0 = all)
	                          ebwt_search.cpp:891-896, 937                                    */
	uint32_t cont_len;     /* -F: length of the reads cut from the FASTA records (< 1024)     */
	uint32_t cont_freq;    /* -F: one read every cont_freq positions                          */
} bt_read_opts;

typedef struct bt_reads bt_reads;
/* spec: comma-separated file names ("-" = stdin, .gz read through zlib, pat.cpp:278-330), or the
 * comma-separated sequences themselves for BT_FMT_CMDLINE ("SEQ" or "SEQ:QUALS") */
int  bt_reads_open(const char* spec, const bt_read_opts* opts, bt_reads** out);
/* parses the next <= max_reads reads with `threads` host threads; the batch (and the name
 * table: names + name_off[n+1]) stays valid until the next call.  n_reads == 0 at the end.  On
 * an input error returns BT_ERR_READS; bt_reads_error() then holds the reference's message. */
int  bt_reads_next(bt_reads* r, uint32_t max_reads, int threads, bt_read_batch* batch,
                   const char** names, const uint64_t** name_off);
/* with BT_READ_KEEP_RAW: the records of the batch last returned, raw_off[n+1] offsets into raw --
 * what the reference dumps for --al / --un / --max (hit.h:385-488) */
int  bt_reads_raw(const bt_reads* r, const char** raw, const uint64_t** raw_off);
/* BT_FMT_TABBED: how many reads of the batch last returned came from records with a second end */
uint32_t bt_reads_paired_count(const bt_reads* r);
const char* bt_reads_error(const bt_reads* r);
void bt_reads_close(bt_reads* r);

typedef struct bt_out_opts {
	int32_t  sam;            /* -S: SAMHitSink (sam.cpp) instead of VerboseHitSink (hit.cpp)   */
	int32_t  full_ref;       /* --fullref: reference names not cut at the first whitespace     */
	int32_t  ref_idx;        /* --refidx: print the reference's index, not its name            */
	int32_t  off_base;       /* -B: added to offsets in the default format                     */
	int32_t  print_cost;     /* --cost: stratum and cost columns (default format)              */
	int32_t  show_seed;      /* --showseed: the read's seed column (default format)            */
	int32_t  mapq;           /* --mapq (255)                                                   */
	int32_t  no_qname_trunc; /* --sam-no-qname-trunc                                           */
	int32_t  no_unal;        /* --no-unal                                                      */
	int32_t  sam_nosq;       /* --sam-nosq (header only)                                       */
	uint32_t khits, mhits;   /* -k / -m: finishRead's rules (hit.h:741-786)                    */
	int32_t  all_hits;       /* -a                                                             */
	int32_t  sample_max;     /* -M: a read over mhits prints one of its buffered best-stratum hits,
	                            picked with the read's seed (hit.cpp:16-68, sam.cpp:263-311)     */
	uint64_t suppress;       /* --suppress: bit f set = 1-based column f+1 of the default
	                            format is left out (hit.cpp:94-297)                            */
} bt_out_opts;

/* HitSink::finish's counters (hit.h:270-346) */
typedef struct bt_out_tally { uint64_t aligned, unaligned, maxed, reported, sample_max, reported_paired; } bt_out_tally;

/* text of all reads of the batch, in read order; *text is malloc'ed (bt_text_free).  tally
 * (optional) is added to. */
int  bt_format_hits(const bt_read_batch* reads, const char* names, const uint64_t* name_off,
                    const bt_hit_batch* hits, const char* const* refnames, const uint32_t* reflens,
                    uint32_t n_refs, const bt_out_opts* o, char** text, size_t* text_len,
                    bt_out_tally* tally);
/* the same for pairs (bt_align_pairs' hit layout): two records per reported pair, upstream mate
 * first; a pair counts once in aligned / unaligned / maxed and twice in reported_paired */
int  bt_format_pairs(const bt_read_batch* r1, const char* names1, const uint64_t* name_off1,
                     const bt_read_batch* r2, const char* names2, const uint64_t* name_off2,
                     const bt_hit_batch* hits, const char* const* refnames, const uint32_t* reflens,
                     uint32_t n_refs, const bt_out_opts* o, char** text, size_t* text_len, bt_out_tally* tally);
/* SAMHitSink::appendHeaders (sam.cpp:20-49) */
int  bt_format_sam_header(const char* const* refnames, const uint32_t* reflens, uint32_t n_refs,
                          const bt_out_opts* o, const char* cmdline, const char* rgline,
                          char** text, size_t* text_len);
/* HitSink::finish's stderr summary (hit.h:270-346), unpaired; tally->sample_max != 0: the -M wording */
int  bt_format_summary(const bt_out_tally* tally, char** text, size_t* text_len);
void bt_text_free(char* text);

#ifdef __cplusplus
}
#endif
#endif /* BOWTIE_AMD_H_ */
