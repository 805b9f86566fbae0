/*
 * bt_io.cpp -- read input and hit output around the search path (host only).
 *
 * Input follows the reference's two-stage pattern sources: a cheap sequential "light parse" that
 * only finds record boundaries (done here on the reading thread), and the per-record parse that
 * the worker threads do (done here by `threads` host threads over the batch):
 *   FASTQ    pat.cpp:797-975      FASTA   pat.cpp:531-640
 *   raw      pat.cpp:1129-1213    -c      pat.cpp:359-528
 *   qualities qual.h:89-153, qual.cpp:38-50      seeds  pat.cpp:21-57
 * Output follows VerboseHitSink::append (hit.cpp:73-301), SAMHitSink::append / reportUnOrMax /
 * appendHeaders (sam.cpp:20-257) and HitSink::finish (hit.h:270-346).
 */
#include <sys/mman.h>
#include "bt_io.h"

#include <ctype.h>
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <thread>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <functional>

/* ---- batch storage ----------------------------------------------------------------------- */
static void* default_alloc(size_t bytes) { return aligned_alloc(256, (bytes + 255u) & ~(size_t)255u); }
static void* (*g_alloc)(size_t) = default_alloc;
static void (*g_dealloc)(void*) = free;
void bt_io_set_allocator(void* (*alloc)(size_t), void (*dealloc)(void*))
{
	g_alloc = alloc ? alloc : default_alloc; g_dealloc = dealloc ? dealloc : free;
}

BtHostBatch::~BtHostBatch() { if (block) block_free(block); }

void BtHostBatch::reset(uint32_t n_reads, uint32_t stride_bytes)
{
	n = n_reads; stride = stride_bytes;
	const size_t need = ((size_t)n_reads * stride_bytes + 255u) & ~(size_t)255u;
	if (need > cap_bytes || n_reads > cap_reads || !block) {
		if (block) block_free(block);
		block = nullptr;
		size_t rows = need > cap_bytes ? need + need / 4 + 256 : cap_bytes;
		rows = (rows + 255u) & ~(size_t)255u;
		uint32_t reads = n_reads > cap_reads ? n_reads + n_reads / 4u + 64u : cap_reads;
		const size_t lens = ((size_t)reads * 2u + 255u) & ~(size_t)255u, seeds = ((size_t)reads * 4u + 255u) & ~(size_t)255u;
		void* (*a)(size_t) = g_alloc; void (*d)(void*) = g_dealloc;
		const size_t total = 2u * rows + lens + seeds + 256u;          /* never an empty request */
		void* b = a(total);
		if (!b && a != default_alloc) { a = default_alloc; d = free; b = a(total); }   /* no pinned memory left: plain memory */
		if (!b) throw std::bad_alloc();
		block = b; block_free = d; cap_bytes = rows; cap_reads = reads;
		seq = (uint8_t*)b; qual = seq + rows;
		len.p = (uint16_t*)(qual + rows); seed.p = (uint32_t*)(qual + rows + lens);
	}
	len.n = n_reads; seed.n = n_reads;
	memset(len.p, 0, (size_t)n_reads * 2u); memset(seed.p, 0, (size_t)n_reads * 4u);
}

static void compact_batch(BtHostBatch* b, const std::vector<uint32_t>& keep)
{
	std::string names, raw;
	std::vector<uint64_t> noff(keep.size() + 1), roff;
	const bool has_raw = b->raw_off.size() == (size_t)b->n + 1;
	const bool has_paired = b->paired.size() == (size_t)b->n;      /* --12: which records had a second end */
	if (has_raw) roff.resize(keep.size() + 1);
	uint32_t np = 0;
	for (size_t k = 0; k < keep.size(); k++) {
		const uint32_t i = keep[k];
		if (i != k) {
			memmove(b->seq + k * b->stride, b->seq + (size_t)i * b->stride, b->stride);
			memmove(b->qual + k * b->stride, b->qual + (size_t)i * b->stride, b->stride);
			b->len[k] = b->len[i]; b->seed[k] = b->seed[i]; b->rdid[k] = b->rdid[i];
			if (has_paired) b->paired[k] = b->paired[i];
		}
		if (has_paired && b->paired[k]) np++;
		noff[k] = names.size(); names.append(b->names, b->name_off[i], b->name_off[i + 1] - b->name_off[i]);
		if (has_raw) { roff[k] = raw.size(); raw.append(b->raw, b->raw_off[i], b->raw_off[i + 1] - b->raw_off[i]); }
	}
	noff[keep.size()] = names.size();
	if (has_raw) { roff[keep.size()] = raw.size(); b->raw.swap(raw); b->raw_off.swap(roff); }
	b->names.swap(names); b->name_off.swap(noff);
	b->n = (uint32_t)keep.size();
	b->len.resize(b->n); b->seed.resize(b->n); b->rdid.resize(b->n);
	b->first_rdid = b->n ? b->rdid[0] : 0;
	if (has_paired) { b->paired.resize(b->n); b->n_paired = np; }
	else if (b->n_paired > b->n) b->n_paired = b->n;
}

bool bt_io_split_tabbed(BtHostBatch* a, BtHostBatch* b, BtHostBatch* unp, std::vector<uint8_t>* order)
{
	order->assign(a->n, 1);
	unp->n = 0; unp->n_paired = 0; unp->paired.clear();
	if (!a->paired.empty() && a->paired.size() != a->n) return false;      /* the column did not follow its batch: never guess which records were pairs */
	if (a->paired.empty() || a->n_paired == a->n) { a->paired.clear(); b->paired.clear(); return true; }       /* pairs only */
	std::vector<uint32_t> kp, ku;
	for (uint32_t i = 0; i < a->n; i++) { if (a->paired[i]) kp.push_back(i); else { ku.push_back(i); (*order)[i] = 0; } }
	/* the unpaired reads: rows of `a` as they are (names and seeds were left alone by the mate-name fix) */
	unp->reset((uint32_t)ku.size(), a->stride);
	unp->rdid.resize(ku.size()); unp->name_off.assign(ku.size() + 1, 0); unp->names.clear();
	const bool has_raw = a->raw_off.size() == (size_t)a->n + 1;
	unp->raw.clear(); unp->raw_off.clear();
	if (has_raw) unp->raw_off.resize(ku.size() + 1);
	for (size_t k = 0; k < ku.size(); k++) {
		const uint32_t i = ku[k];
		memcpy(unp->seq + k * unp->stride, a->seq + (size_t)i * a->stride, a->stride);
		memcpy(unp->qual + k * unp->stride, a->qual + (size_t)i * a->stride, a->stride);
		unp->len[k] = a->len[i]; unp->seed[k] = a->seed[i]; unp->rdid[k] = a->rdid[i];
		unp->name_off[k] = unp->names.size(); unp->names.append(a->names, a->name_off[i], a->name_off[i + 1] - a->name_off[i]);
		if (has_raw) { unp->raw_off[k] = unp->raw.size(); unp->raw.append(a->raw, a->raw_off[i], a->raw_off[i + 1] - a->raw_off[i]); }
	}
	unp->name_off[ku.size()] = unp->names.size();
	if (has_raw) unp->raw_off[ku.size()] = unp->raw.size();
	unp->first_rdid = unp->n ? unp->rdid[0] : 0; unp->end_rdid = a->end_rdid;
	a->paired.clear(); b->paired.clear();
	compact_batch(a, kp); compact_batch(b, kp);
	a->n_paired = a->n; b->n_paired = b->n;
	return true;
}

bool bt_io_intersect_pairs(BtHostBatch* a, BtHostBatch* b)
{
	const bool same_end = a->end_rdid == b->end_rdid;
	bool equal = a->n == b->n;
	for (uint32_t i = 0; equal && i < a->n; i++) equal = a->rdid[i] == b->rdid[i];
	if (equal) return same_end;
	std::vector<uint32_t> ka, kb;
	uint32_t i = 0, k = 0;
	while (i < a->n && k < b->n) {
		if (a->rdid[i] == b->rdid[k]) { ka.push_back(i++); kb.push_back(k++); }
		else if (a->rdid[i] < b->rdid[k]) i++; else k++;
	}
	compact_batch(a, ka); compact_batch(b, kb);
	return same_end;
}

bt_read_batch BtHostBatch::view() const
{
	bt_read_batch b;
	b.n_reads = n; b.stride = stride; b.seq = seq; b.qual = qual; b.len = len.data(); b.seed = seed.data();
	return b;
}

/* ---- the stream -------------------------------------------------------------------------- */
struct BtRec { size_t off; uint32_t len; uint64_t rdid; };

/* The reader's own large blocks (the file window: a batch's raw text, 2.9 GB for 12 M reads; the newline index) are asked to be
 * backed by transparent huge pages where the host offers them on request: a fault and an unmap per 2 MB instead of per 4 KB
 * (round 6, GPU call 9: 16 GB first touched by 64 threads in 0.17 s instead of 1.6 s, unmapped in 0.7 s instead of 1.6 s).
 * Only with BT_IO_HUGEPAGES=1: measured in the binary (GPU call 11) it costs more than it saves -- the window grows fill by fill,
 * every growth is an mremap() and a madvise() under the address space's write lock, and the threads loading the index beside it
 * fault under the read lock: the first batch was submitted 0.66 s later (1.96 s instead of 1.30 s) for 0.05 s per batch saved. */
static void advise_huge(void* p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
	static const bool on = getenv("BT_IO_HUGEPAGES") && atoi(getenv("BT_IO_HUGEPAGES")) != 0;
	if (!on || !p || bytes < ((size_t)4u << 20)) return;
	const uintptr_t a = ((uintptr_t)p + 4095u) & ~(uintptr_t)4095u, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095u;
	if (e > a) (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
#else
	(void)p; (void)bytes;
#endif
}

/* f(0) .. f(T - 1) side by side, on threads that are started once and kept.  Until round 6's call 13 every phase of the reader
 * started T threads of its own and joined them: on the GPU box's host (256 hardware threads) that is ~0.8 ms per thread -- a
 * stack mapped and unmapped again, with a TLB shoot-down across the machine -- 0.05 s per phase at -p 64, five phases per
 * batch.  One job at a time: a second caller (another stream read from another thread), or a process that forked after the
 * pool was made, starts threads of its own as before.  BT_IO_POOL=0: always. */
namespace {
struct IoPool {
	std::mutex job_m;
	std::mutex m; std::condition_variable cv_work, cv_done;
	std::vector<std::thread> th;
	bool stop = false;
	uint64_t gen = 0; int count = 0; std::atomic<int> next{0}; size_t active = 0;
	const std::function<void(int)>* fn = nullptr;
	const pid_t pid = getpid();
	void loop(uint64_t seen)
	{
		for (;;) {
			std::unique_lock<std::mutex> l(m);
			cv_work.wait(l, [&] { return stop || gen != seen; });
			if (stop) return;
			seen = gen;
			l.unlock();
			for (;;) { const int i = next.fetch_add(1); if (i >= count) break; (*fn)(i); }
			l.lock();
			if (--active == 0) cv_done.notify_all();
		}
	}
	void run(int T, const std::function<void(int)>& f)
	{
		{
			std::lock_guard<std::mutex> l(m);
			while ((int)th.size() < T - 1) { const uint64_t g = gen; th.emplace_back([this, g] { loop(g); }); }
			fn = &f; count = T; next.store(0); active = th.size(); gen++;
		}
		cv_work.notify_all();
		for (;;) { const int i = next.fetch_add(1); if (i >= T) break; f(i); }      /* the caller takes its share */
		std::unique_lock<std::mutex> l(m);
		cv_done.wait(l, [&] { return active == 0; });
	}
};
}
static void par_for(int T, const std::function<void(int)>& f)
{
	if (T <= 1) { f(0); return; }
	static const bool use_pool = !(getenv("BT_IO_POOL") && atoi(getenv("BT_IO_POOL")) == 0);
	if (use_pool) {
		static IoPool* const pool = new IoPool();      /* never taken down: its idle threads go with the process, whichever way it leaves */
		if (getpid() == pool->pid && pool->job_m.try_lock()) {
			std::lock_guard<std::mutex> g(pool->job_m, std::adopt_lock);
			pool->run(T, f);
			return;
		}
	}
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++) th.emplace_back(f, t);
	for (auto& x : th) x.join();
}

/* the file window: grows without being zero-filled (realloc moves big blocks by remapping, not by copying) */
struct BtWindow {
	char* p = nullptr; size_t n = 0;
	BtWindow() {}
	~BtWindow() { free(p); }
	BtWindow(const BtWindow&) = delete;
	BtWindow& operator=(const BtWindow&) = delete;
	char* data() { return p; }
	const char* data() const { return p; }
	size_t size() const { return n; }
	char& operator[](size_t i) { return p[i]; }
	void resize(size_t m) { char* q = (char*)realloc(p, m ? m : 1); if (!q) throw std::bad_alloc(); if (q != p || m > n) advise_huge(q, m); p = q; n = m; }
};

/* offsets, appended in bulk by several threads (no zero-fill on growth) */
struct BtOffsets {
	size_t* p = nullptr; size_t n = 0, cap = 0;
	BtOffsets() {}
	~BtOffsets() { free(p); }
	BtOffsets(const BtOffsets&) = delete;
	BtOffsets& operator=(const BtOffsets&) = delete;
	void reserve(size_t m) { if (m > cap) { const size_t c = m + m / 4 + 1024; size_t* q = (size_t*)realloc(p, c * sizeof(size_t)); if (!q) throw std::bad_alloc(); advise_huge(q, c * sizeof(size_t)); p = q; cap = c; } }
	size_t size() const { return n; }
	size_t& operator[](size_t i) { return p[i]; }
	void resize(size_t m) { reserve(m); n = m; }
};

struct BtReadStream {
	bt_read_opts o;
	std::vector<std::string> items;       /* file names, or the -c sequences                     */
	size_t item = 0;
	gzFile f = nullptr;
	bool file_first = true;               /* nothing of the current file consumed yet            */
	BtWindow buf;                         /* file window                                         */
	size_t pos = 0, end = 0;
	bool feof = false;
	uint64_t file_recs = 0;               /* complete records seen in the current file           */
	uint64_t rdid = 0;                    /* next read id (counts skipped reads too)             */
	uint64_t limit = ~0ull;               /* first read id that is not processed                 */
	bool done = false;
	bool open_failed = false;             /* the files left could not be opened: the run fails (CFilePatternSource::open
	                                         throws, pat.cpp:296-357) once the reads before them are out            */
	std::string raw;                      /* the batch's record texts                            */
	/* the FASTQ bulk path (next_fastq): an uncompressed regular file is read straight from its descriptor -- several threads,
	 * one pread each, into the window -- instead of through zlib's copy; and the window's newlines are indexed, by the same
	 * threads, as the data arrives: the light parse (four newlines to a record) then walks the index instead of calling
	 * memchr four times per record (the input stage was one thread's scan, 10-11 M reads/s whatever -p said) */
	int rawfd = -1; uint64_t rawoff = 0;
	BtOffsets nl;                         /* offsets into the window of the '\n's in [.., idx_end), ascending */
	size_t nl_cur = 0, idx_end = 0;
	int io_threads = 1;
	double avg_rec = 256.0;               /* bytes per record of the last batch: how far ahead of a batch's need to read */
	/* -F: the sliding window over the FASTA text (FastaContinuousPatternSource's members) */
	size_t c_eat = 0, c_bufcur = 0; bool c_begin = true; uint64_t c_cur = 0, c_last = 0;
	char c_buf[1024]; std::string c_prefix;
	std::vector<BtRec> recs;
};

static void st_close_file(BtReadStream* s)
{
	if (s->f) { gzclose(s->f); s->f = nullptr; }
	if (s->rawfd >= 0) { close(s->rawfd); s->rawfd = -1; }
}

static int st_fill(BtReadStream* s)
{
	/* keep the unread tail, top the window up */
	if (s->pos > 0) {
		memmove(s->buf.data(), s->buf.data() + s->pos, s->end - s->pos);
		s->end -= s->pos; s->pos = 0;
	}
	if (s->feof || !s->f) return 0;
	const size_t room = s->buf.size() - s->end;
	const int got = gzread(s->f, s->buf.data() + s->end, (unsigned)room);
	if (got <= 0) { s->feof = true; return 0; }
	s->end += (size_t)got;
	return got;
}
static inline int st_getc(BtReadStream* s)
{
	if (s->pos == s->end) { if (st_fill(s) == 0) return -1; }
	return (unsigned char)s->buf[s->pos++];
}
static inline int st_peek(BtReadStream* s)
{
	if (s->pos == s->end) { if (st_fill(s) == 0) return -1; }
	return (unsigned char)s->buf[s->pos];
}

static bool st_open_next(BtReadStream* s, std::string* err, bool keep_window = false)
{
	st_close_file(s);
	if (s->item >= s->items.size()) return false;
	const std::string& fn = s->items[s->item++];
	/* a name starting with \x01: a file the caller has already taken out of the run and said so (bowtie-amd: its -Q
	 * partner could not be opened) -- it counts as one that does not open, silently */
	const bool taken_out = !fn.empty() && fn[0] == '\x01';
	s->f = taken_out ? nullptr : (fn == "-") ? gzdopen(0, "rb") : gzopen(fn.c_str(), "rb");
	if (!s->f) {
		/* the reference warns and moves on to the next file (pat.cpp:301-319); if none of the files left opens, the
		 * run fails (:357) */
		if (!taken_out) {
			*err = "Warning: Could not open read file \"" + fn + "\" for reading; skipping...";
			fprintf(stderr, "%s\n", err->c_str());
			err->clear();
		}
		if (s->item >= s->items.size()) { s->open_failed = true; return false; }
		return st_open_next(s, err, keep_window);
	}
	gzbuffer(s->f, 1u << 20);
	s->rawoff = 0;
	if (fn != "-" && s->o.format == BT_FMT_FASTQ && !getenv("BT_IO_NO_RAW")) {
		/* a regular file that is not gzip's: the bulk path reads it from a descriptor of its own (fq_more) */
		const int fd = open(fn.c_str(), O_RDONLY | O_CLOEXEC);
		if (fd >= 0) {
			struct stat sb; unsigned char m[2] = {0, 0};
			const bool plain = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && !(pread(fd, m, 2, 0) == 2 && m[0] == 0x1f && m[1] == 0x8b);
			if (plain) s->rawfd = fd; else close(fd);
		}
	}
	s->file_first = true; s->feof = false; s->file_recs = 0;
	if (!keep_window) s->pos = s->end = 0;
	return true;
}

BtReadStream* bt_io_open(const char* spec, const bt_read_opts& opts, std::string* err)
{
	BtReadStream* s = new BtReadStream();
	s->o = opts;
	const char* p = spec ? spec : "";
	while (*p) {                               /* tokenize(..., ",") : empty tokens vanish */
		const char* q = strchr(p, ',');
		const size_t n = q ? (size_t)(q - p) : strlen(p);
		if (n) s->items.emplace_back(p, n);
		p += n + (q ? 1 : 0);
	}
	s->buf.resize(8u << 20);
	if (opts.upto != 0 && opts.skip + opts.upto > opts.skip) s->limit = opts.skip + opts.upto;
	(void)err;
	return s;
}

void bt_io_close(BtReadStream* s)
{
	if (!s) return;
	st_close_file(s);
	delete s;
}

/* ---- light parse: one record's text appended to s->raw ------------------------------------- */
/* Each returns 1 = a record was appended, 0 = the current file is finished, -1 = error (light_fastq: -2 = the file
 * ended inside a record). */

static int light_fastq(BtReadStream* s, std::string* err)
{
	std::string& raw = s->raw;
	const size_t start = raw.size();
	if (s->file_first) {
		int c = st_getc(s);
		while (c == '\r' || c == '\n') c = st_getc(s);
		if (c != '@') {
			/* the reference checks the first character only (an empty file fails the same way) */
			*err = "Error: reads file does not look like a FASTQ file";
			return -1;
		}
		s->file_first = false;
		raw.push_back('@');
	}
	int newlines = 4;
	while (newlines) {
		/* bulk path: copy up to the next '\n' in the window */
		if (s->pos < s->end) {
			const char* b = s->buf.data() + s->pos;
			const char* nl = (const char*)memchr(b, '\n', s->end - s->pos);
			const size_t take = nl ? (size_t)(nl - b) + 1 : s->end - s->pos;
			raw.append(b, take);
			s->pos += take;
			if (nl) newlines--;
			continue;
		}
		const int c = st_getc(s);
		if (c < 0) {
			if (newlines == 1) { raw.push_back('\n'); newlines = 0; break; }   /* EOF stands in for the last newline */
			/* clean end of file (0), or EOF inside a record (-2): the partial record is dropped, and the caller gives
			 * up the record before it as well, as the reference's light parser does (pat.cpp:826-856: `aborted`) */
			const bool clean = newlines == 4;          /* no line of a record was complete yet (its characters are lost all the same) */
			raw.resize(start);
			return clean ? 0 : -2;
		}
		s->pos--;            /* window refilled: take the bulk path */
	}
	return 1;
}

static int light_fasta(BtReadStream* s, std::string* err)
{
	std::string& raw = s->raw;
	if (s->file_first) {
		int c = st_getc(s);
		if (c < 0) return 0;
		while (c == '\r' || c == '\n') c = st_getc(s);
		if (c != '>') { *err = "Error: reads file does not look like a FASTA file"; return -1; }
		s->file_first = false;
	} else if (s->feof && s->pos == s->end) return 0;
	const size_t start = raw.size();
	raw.push_back('>');
	for (;;) {
		if (s->pos == s->end && st_fill(s) == 0) break;
		const char* b = s->buf.data() + s->pos;
		const char* gt = (const char*)memchr(b, '>', s->end - s->pos);
		const size_t take = gt ? (size_t)(gt - b) : s->end - s->pos;
		raw.append(b, take);
		s->pos += take;
		if (gt) { s->pos++; return 1; }          /* the next record's '>' is consumed here */
	}
	/* EOF: a lone '>' is no record */
	if (raw.size() == start + 1) { raw.resize(start); return 0; }
	return 1;
}

static int light_raw(BtReadStream* s, std::string* err)
{
	(void)err;
	std::string& raw = s->raw;
	s->file_first = false;
	int c = st_getc(s);
	while (c == '\n' || c == '\r') c = st_getc(s);
	if (c < 0) return 0;
	while (c >= 0 && c != '\n' && c != '\r') { raw.push_back((char)c); c = st_getc(s); }
	if (c == '\n') {
		raw.push_back('\n');
		if (st_peek(s) == '\r') { raw.push_back('\r'); s->pos++; }
	}
	return 1;
}

/* -F <len>,<freq>: every len-mer at interval freq of each FASTA record, named <record name>_<offset>
 * (pat.cpp:651-724).  The sequence name stops at its first whitespace; characters that are no
 * nucleotide code are skipped, ambiguity codes and '-' become N. */
static int dna_cat(int c)
{
	if (c < 0) return 0;
	if (strchr("ACGTacgt", c) && c) return 1;
	if (strchr("BDHKMNRSVWXYbdhkmnrsvwxy", c) && c) return 2;
	return c == '-' ? 3 : 0;
}
static int light_fasta_cont(BtReadStream* s, std::string* err)
{
	(void)err;
	const size_t length = s->o.cont_len, freq = s->o.cont_freq;
	s->file_first = false;
	for (;;) {
		int c = st_getc(s);
		if (c < 0) return 0;
		if (c == '>') {
			s->c_eat = length - 1; s->c_prefix.clear(); s->c_begin = true; s->c_bufcur = 0; s->c_last = s->c_cur;
			c = st_getc(s);
			bool sawSpace = false;
			while (c >= 0 && c != '\n' && c != '\r') {
				if (!sawSpace) sawSpace = isspace(c) != 0;
				if (!sawSpace) s->c_prefix.push_back((char)c);
				c = st_getc(s);
			}
			while (c == '\n' || c == '\r') c = st_getc(s);
			if (c < 0) return 0;
			s->c_prefix.push_back('_');
		}
		const int cat = dna_cat(c);
		if (cat >= 2) c = 'N';
		if (cat == 0) continue;
		s->c_buf[s->c_bufcur++] = (char)c;
		if (s->c_bufcur == 1024) s->c_bufcur = 0;
		if (s->c_eat > 0) {
			s->c_eat--;
			if (!s->c_begin) s->c_cur++;
			continue;
		}
		char nb[24]; snprintf(nb, sizeof(nb), "%llu", (unsigned long long)(s->c_cur - s->c_last));
		s->raw.append(s->c_prefix); s->raw.append(nb); s->raw.push_back('\t');
		for (size_t i = 0; i < length; i++) {
			const size_t back = length - i;
			s->raw.push_back(back <= s->c_bufcur ? s->c_buf[s->c_bufcur - back] : s->c_buf[s->c_bufcur + 1024 - back]);
		}
		s->c_eat = freq - 1;
		s->c_cur++;
		s->c_begin = false;
		return 1;
	}
}

/* ---- per-record parse ---------------------------------------------------------------------- */
struct BtParsed {
	std::string seq, qual;      /* codes 0..4, Phred+33 */
	size_t name_b = 0, name_n = 0;   /* name = rec[name_b, name_b + name_n), or the read id when empty */
	bool ok = true;             /* false: the record ended prematurely -- the reference skips it */
	bool paired = false;        /* --12: the record had a second end */
	bool name_as_is = false;    /* --12: an empty name stays empty (the other readers put the read id there) */
};

static const uint8_t* asc2dna_table()
{
	/* built once, by whichever thread gets here first (C++11 guarantees the initialisation of a
	 * function-local static is race-free); the parse threads only ever read it */
	struct Table { uint8_t t[256]; Table() { memset(t, 4, sizeof(t)); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; } };
	static const Table tab;
	return tab.t;
}

/* solexaToPhred (qual.h:29-33): round(10 log10(10^(sol/10) + 1)) */
static int solexa_to_phred(int sol)
{
	if (sol < -10) return 0;
	return (int)(10.0 * log(1.0 + pow(10.0, sol / 10.0)) / log(10.0) + 0.5);
}

static bool qual_to_phred33(int c, int enc, char* out, std::string* err)
{
	char tmp[256];
	if (c == ' ') {
		*err = "Saw a space but expected an ASCII-encoded quality value.\n"
		       "Are quality values formatted as integers?  If so, try --integer-quals.";
		return false;
	}
	if (enc == BT_QUAL_SOLEXA64) {
		const int cc = (int)(char)(solexa_to_phred(c - 64) + 33);
		if (cc < 33) {
			snprintf(tmp, sizeof(tmp), "Saw ASCII character %d but expected 64-based Solexa qual (converts to %d).\n"
			         "Try not specifying --solexa-quals.", c, cc);
			*err = tmp; return false;
		}
		c = cc;
	} else if (enc == BT_QUAL_PHRED64) {
		if (c < 64) {
			snprintf(tmp, sizeof(tmp), "Saw ASCII character %d but expected 64-based Phred qual.\n"
			         "Try not specifying --solexa1.3-quals/--phred64-quals.", c);
			*err = tmp; return false;
		}
		c -= (64 - 33);
	} else if (c < 33) {
		snprintf(tmp, sizeof(tmp), "Saw ASCII character %d but expected 33-based Phred qual.", c);
		*err = tmp; return false;
	}
	*out = (char)c;
	return true;
}

static void trim_end(std::string& s, size_t n) { s.resize(n >= s.size() ? 0 : s.size() - n); }

static std::string name_of(const char* rec, const BtParsed& p, uint64_t rdid)
{
	if (p.name_n || p.name_as_is) return std::string(rec + p.name_b, p.name_n);
	char b[24]; snprintf(b, sizeof(b), "%llu", (unsigned long long)rdid);
	return b;
}

static bool parse_fastq(const char* r, size_t n, const bt_read_opts& o, uint64_t rdid, BtParsed* p, std::string* err)
{
	const uint8_t* a2d = asc2dna_table();
	size_t cur = 1;
	int c = '\n';
#define NEXT() ((cur < n) ? (unsigned char)r[cur++] : (cur++, '\n'))
	p->name_b = 1;
	for (;;) {
		c = NEXT();
		if (c == '\n' || c == '\r') {
			do { c = NEXT(); } while ((c == '\n' || c == '\r') && cur < n);
			break;
		}
		p->name_n++;
	}
	int nchar = 0;
	while (c != '+' && cur < n) {
		if (c == '.') c = 'N';
		if (isalpha(c)) { if (nchar++ >= o.trim5) p->seq.push_back((char)a2d[c]); }
		c = NEXT();
	}
	const size_t trimmed5 = (size_t)nchar - p->seq.size();
	const size_t before = p->seq.size();
	trim_end(p->seq, (size_t)o.trim3);
	const size_t trimmed3 = before - p->seq.size();
	if (c != '+') { *err = "Error: reads file does not look like a FASTQ file"; return false; }
	do { c = NEXT(); } while (c != '\n' && c != '\r' && cur <= n);
	while (cur < n && (c == '\n' || c == '\r')) c = NEXT();
	size_t nqual = 0;
	char q;
	if (o.qual_enc == BT_QUAL_INT || o.qual_enc == BT_QUAL_INT_SOLEXA) {
		/* space-separated integers (pat.cpp:918-936); the reference neither trims these at the 3' end
		 * nor compares their number with the bases -- here a mismatch in number is an error */
		int cur_int = 0;
		while (c != '\t' && c != '\n' && c != '\r') {
			cur_int = cur_int * 10 + (c - '0');
			c = NEXT();
			if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
				int pq = (o.qual_enc == BT_QUAL_INT_SOLEXA) ? solexa_to_phred(cur_int) + 33 : (cur_int <= 93 ? cur_int : 93) + 33;
				if (pq < 33) { char b[96]; snprintf(b, sizeof(b), "Saw negative Phred quality %d.", pq - 33); *err = b; return false; }
				cur_int = 0;
				if (c == ' ') c = NEXT();
				if (++nqual > (size_t)o.trim5) p->qual.push_back((char)pq);
			}
			if (cur > n + 1) break;
		}
		trim_end(p->qual, trimmed3);
		if (p->qual.size() < p->seq.size()) {
			*err = "Too few quality values for read: " + name_of(r, *p, rdid) + "\n\tare you sure this is a FASTQ-int file?";
			return false;
		}
		if (p->qual.size() > p->seq.size()) {
			*err = "Reads file contained a pattern with more than 1024 quality values.\n"
			       "Please truncate reads and quality values and and re-run Bowtie";
			return false;
		}
		return true;
	}
	if (!qual_to_phred33(c, o.qual_enc, &q, err)) return false;
	if (nqual++ >= trimmed5) p->qual.push_back(q);
	while (cur < n) {
		c = NEXT();
		if (c == ' ') {
			*err = "Encountered a space parsing the quality string for read " + name_of(r, *p, rdid) + "\n"
			       "If this is a FASTQ file with integer (non-ASCII-encoded) qualities, please\n"
			       "re-run Bowtie with the --integer-quals option.";
			return false;
		}
		if (c == '\r' || c == '\n') break;
		if (!qual_to_phred33(c, o.qual_enc, &q, err)) return false;
		if (nqual++ >= trimmed5) p->qual.push_back(q);
	}
	trim_end(p->qual, trimmed3);
	if (p->qual.size() < p->seq.size()) {
		*err = "Too few quality values for read: " + name_of(r, *p, rdid) + "\n\tare you sure this is a FASTQ-int file?";
		return false;
	}
	if (p->qual.size() > p->seq.size()) {
		*err = "Reads file contained a pattern with more than 1024 quality values.\n"
		       "Please truncate reads and quality values and and re-run Bowtie";
		return false;
	}
#undef NEXT
	return true;
}

static bool parse_fasta(const char* r, size_t n, const bt_read_opts& o, BtParsed* p)
{
	const uint8_t* a2d = asc2dna_table();
	size_t cur = 1;
	int c = -1;
	p->name_b = 1;
	while (cur < n) {
		c = (unsigned char)r[cur++];
		if (c == '\n' || c == '\r') {
			do { c = (cur < n) ? (unsigned char)r[cur] : '\n'; cur++; } while ((c == '\n' || c == '\r') && cur < n);
			break;
		}
		p->name_n++;
	}
	if (cur >= n) { p->ok = false; return true; }          /* "FASTA ended prematurely": the read is skipped */
	int nchar = 0;
	/* the first sequence line only; a final character that sits at the very end of the record
	 * (no newline before EOF) is not consumed -- as in the reference */
	while (c != '\n' && cur < n) {
		if (c == '.') c = 'N';
		if (isalpha(c)) { if (nchar++ >= o.trim5) p->seq.push_back((char)a2d[c]); }
		c = (unsigned char)r[cur++];
	}
	trim_end(p->seq, (size_t)o.trim3);
	p->qual.assign(p->seq.size(), 'I');
	return true;
}

static bool parse_raw(const char* r, size_t n, const bt_read_opts& o, BtParsed* p)
{
	const uint8_t* a2d = asc2dna_table();
	int nchar = 0;
	for (size_t cur = 0; cur < n; cur++) {
		const int c = (unsigned char)r[cur];
		if (isalpha(c)) { if (nchar++ >= o.trim5) p->seq.push_back((char)a2d[c]); }
	}
	trim_end(p->seq, (size_t)o.trim3);
	p->qual.assign(p->seq.size(), 'I');
	p->name_n = 0;                                        /* name = read id */
	return true;
}

/* --12 (TabbedPatternSource::parse, pat.cpp:1017-1127): "name<TAB>seq<TAB>quals", optionally followed by
 * "<TAB>seq2<TAB>quals2" -- the second end of a pair, under the same name.  A line that stops before the
 * qualities of an end is dropped whole.  Sequence characters that are no letters are skipped; the number of
 * qualities is compared with the number of bases before either is trimmed.  Both ends are parsed (their errors
 * are the record's); the one the stream was opened for is kept.
 * A close restatement of TabbedPatternSource::parse (pat.cpp:1017-1127), loop for loop, kept that way on purpose: what
 * malformed lines do (which are dropped, which are errors, with which message) follows from the order of its tests,
 * and the differential parser fuzz (tests/test_parser_fuzz.py) holds it to that. */
static bool parse_tabbed(const char* r, size_t n, const bt_read_opts& o, uint64_t rdid, BtParsed* p, std::string* err)
{
	const uint8_t* a2d = asc2dna_table();
	const bool want2 = (o.flags & BT_READ_MATE2) != 0;
	size_t cur = 0;
	int c = (unsigned char)r[cur++];
	p->name_b = 0; p->name_n = 0; p->name_as_is = true;
	while (c != '\t' && cur < n) { p->name_n++; c = (unsigned char)r[cur++]; }
	if (cur >= n) { p->ok = false; return true; }
	for (int e = 0; e < 2 && c == '\t'; e++) {
		std::string seq, qual;
		int nchar = 0, nqual = 0;
		c = (unsigned char)r[cur++];
		while (c != '\t' && cur < n) {
			if (isalpha(c)) { if (nchar++ >= o.trim5) seq.push_back((char)a2d[c]); }
			c = (unsigned char)r[cur++];
		}
		if (cur >= n) { p->seq.clear(); p->qual.clear(); p->ok = false; return true; }
		trim_end(seq, (size_t)o.trim3);
		c = (unsigned char)r[cur++];
		while (c != '\t' && c != '\n' && c != '\r') {
			if (c == ' ') {
				*err = "Encountered a space parsing the quality string for read " + name_of(r, *p, rdid) + "\n"
				       "If this is a FASTQ file with integer (non-ASCII-encoded) qualities, please\n"
				       "re-run Bowtie with the --integer-quals option.";
				return false;
			}
			char q;
			if (!qual_to_phred33(c, o.qual_enc, &q, err)) return false;
			if (++nqual > o.trim5) qual.push_back(q);
			if (cur >= n) break;
			c = (unsigned char)r[cur++];
		}
		if (nchar > nqual) {
			*err = "Too few quality values for read: " + name_of(r, *p, rdid) + "\n\tare you sure this is a FASTQ-int file?";
			return false;
		}
		if (nqual > nchar) {
			*err = "Reads file contained a pattern with more than 1024 quality values.\n"
			       "Please truncate reads and quality values and and re-run Bowtie";
			return false;
		}
		trim_end(qual, (size_t)o.trim3);
		if (e == 1) p->paired = true;
		if ((e == 1) == want2) { p->seq.swap(seq); p->qual.swap(qual); }
	}
	return true;
}

static bool parse_fasta_cont(const char* r, size_t n, const bt_read_opts& o, BtParsed* p)
{
	const uint8_t* a2d = asc2dna_table();
	const char* tab = (const char*)memchr(r, '\t', n);
	if (!tab || (size_t)(tab - r) + 1 >= n) { p->ok = false; return true; }
	p->name_b = 0; p->name_n = (size_t)(tab - r);
	/* -5 / -3 do not reach this source: the reference constructs it with both trims 0 (pat.h:600-604) */
	(void)o;
	for (size_t cur = (size_t)(tab - r) + 1; cur < n; cur++) {
		const int c = (unsigned char)r[cur];
		if (isalpha(c)) p->seq.push_back((char)a2d[c]);
	}
	p->qual.assign(p->seq.size(), 'I');
	return true;
}

/* -c: "SEQ" or "SEQ:QUALS"; qualities are always Phred+33 here (pat.cpp:502) */
static bool parse_cmdline(const char* r, size_t n, const bt_read_opts& o, uint64_t rdid, BtParsed* p, std::string* err)
{
	const uint8_t* a2d = asc2dna_table();
	const char* colon = (const char*)memchr(r, ':', n);
	const size_t sl = colon ? (size_t)(colon - r) : n;
	if (sl == 0) { p->ok = false; return true; }
	int nchar = 0;
	/* the token is walked up to, not including, its last character's successor; the reference's
	 * buffer always has a tab after the sequence, so every character is seen */
	for (size_t i = 0; i < sl; i++) {
		const int c = (unsigned char)r[i];
		if (isalpha(c)) { if (nchar++ >= o.trim5) p->seq.push_back((char)a2d[c]); }
	}
	trim_end(p->seq, (size_t)o.trim3);
	int nqual = 0;
	if (colon) {
		for (size_t i = sl + 1; i < n; i++) {
			const int c = (unsigned char)r[i];
			if (c == '\t' || c == '\n' || c == '\r') break;
			char q;
			if (c == ' ') { *err = "Encountered a space parsing the quality string for read " + name_of(r, *p, rdid); return false; }
			if (!qual_to_phred33(c, BT_QUAL_PHRED33, &q, err)) return false;
			if (++nqual > o.trim5) p->qual.push_back(q);
		}
	} else {
		/* default qualities are one 'I' per character of the token */
		for (size_t i = 0; i < sl; i++) { if (++nqual > o.trim5) p->qual.push_back('I'); }
	}
	if (nchar > nqual) { *err = "Too few quality values for read: " + name_of(r, *p, rdid) + "\n\tare you sure this is a FASTQ-int file?"; return false; }
	if (nqual > nchar) {
		*err = "Reads file contained a pattern with more than 1024 quality values.\n"
		       "Please truncate reads and quality values and and re-run Bowtie";
		return false;
	}
	trim_end(p->qual, (size_t)o.trim3);
	p->name_n = 0;
	return true;
}

/* genRandSeed (pat.cpp:21-57) */
static uint32_t rand_seed(const uint8_t* seq, const uint8_t* qual, size_t len, const char* name, size_t name_n, uint32_t seed)
{
	uint32_t r = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	for (size_t i = 0; i < len; i++) r ^= (uint32_t)seq[i] << ((i & 15u) << 1);
	for (size_t i = 0; i < len; i++) r ^= (uint32_t)qual[i] << ((i & 3u) << 3);
	for (size_t i = 0; i < name_n; i++) r ^= (uint32_t)(int)(signed char)name[i] << ((i & 3u) << 3);
	return r;
}

/* ---- FASTQ, the format that matters for throughput -------------------------------------------
 * The file is read in large pieces into one window; a batch's records are spans of that window
 * (found by counting newlines, the reference's light parse), and the host threads turn them
 * straight into rows of the batch.  A record of the everyday shape -- no '\r', a name, only
 * letters or '.' on the sequence line, a '+' line, as many quality characters as bases, all of
 * them valid for the encoding -- takes a short loop; anything else goes through parse_fastq
 * above, which follows the reference's parser step by step (and produces its error messages). */
struct FqRec { size_t off; uint32_t e[4]; uint64_t rdid; };   /* e[k]: offset of line k's '\n' from off */
/* a vector of plain records that several threads fill in place: growing it does not touch the new elements */
template <class T> struct BtPod {
	T* p = nullptr; size_t n = 0, cap = 0;
	BtPod() {}
	~BtPod() { free(p); }
	BtPod(const BtPod&) = delete;
	BtPod& operator=(const BtPod&) = delete;
	void reserve(size_t m) { if (m > cap) { const size_t c = m + m / 4 + 64; T* q = (T*)realloc((void*)p, c * sizeof(T)); if (!q) throw std::bad_alloc(); advise_huge(q, c * sizeof(T)); p = q; cap = c; } }
	void resize_uninit(size_t m) { reserve(m); n = m; }
	void push_back(const T& v) { reserve(n + 1); p[n++] = v; }
	void pop_back() { n--; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	T* data() { return p; }
	T& operator[](size_t i) { return p[i]; }
	const T& operator[](size_t i) const { return p[i]; }
	T& back() { return p[n - 1]; }
	T& front() { return p[0]; }
};

static double io_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static double g_io_fill_s = 0;        /* BT_IO_PROFILE: time inside gzread / pread */
static double g_io_index_s = 0;       /* BT_IO_PROFILE: time indexing the window's newlines */

/* the '\n's of window[lo, hi): counted, or their offsets written to out[0..) in order (-> how many) */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static size_t nl_scan_avx2(const char* w, size_t lo, size_t hi, size_t* out)
{
	const __m256i nlv = _mm256_set1_epi8('\n');
	size_t i = lo, n = 0;
	for (; i + 64 <= hi; i += 64) {
		const __m256i a = _mm256_loadu_si256((const __m256i*)(w + i)), b = _mm256_loadu_si256((const __m256i*)(w + i + 32));
		uint64_t m = (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nlv)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nlv)) << 32);
		if (!out) { n += (size_t)__builtin_popcountll(m); continue; }
		while (m) { out[n++] = i + (size_t)__builtin_ctzll(m); m &= m - 1; }
	}
	for (; i < hi; i++) if (w[i] == '\n') { if (out) out[n] = i; n++; }
	return n;
}
#endif
static size_t nl_scan(const char* w, size_t lo, size_t hi, size_t* out)
{
#if defined(__x86_64__)
	static const bool avx2 = __builtin_cpu_supports("avx2");
	if (avx2) return nl_scan_avx2(w, lo, hi, out);
#endif
	const char* p = w + lo; const char* e = w + hi;
	size_t n = 0;
	while (p < e) {
		const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
		if (!q) break;
		if (out) out[n] = (size_t)(q - w);
		n++;
		p = q + 1;
	}
	return n;
}
/* tests: BT_IO_FILL_BYTES / BT_IO_SLICE_BYTES shrink a fill of the window and a thread's share of it, so that files of a few
 * KB cross every boundary the 64 MB / 8 MB / 2 MB defaults put in the way of files of many GB */
static size_t io_knob(const char* name, size_t dflt)
{
	const char* e = getenv(name);
	const size_t v = e && *e ? (size_t)strtoull(e, nullptr, 10) : 0;
	return v ? v : dflt;
}
/* extend the newline index over window[idx_end, end): the io threads take a slice each -- count, then write in place */
static void nl_index_more(BtReadStream* s)
{
	const size_t lo = s->idx_end, hi = s->end;
	if (hi <= lo) return;
	const double t0 = io_now();
	const char* w = s->buf.data();
	const size_t span = hi - lo;
	static const size_t sliceMin = io_knob("BT_IO_SLICE_BYTES", 2u << 20);
	int T = s->io_threads < 1 ? 1 : s->io_threads;
	if (span < 2u * sliceMin) T = 1;
	else if ((size_t)T > span / sliceMin) T = (int)(span / sliceMin);
	std::vector<size_t> cnt((size_t)T + 1, 0);
	auto slice = [&](int t, size_t* a, size_t* b) { *a = lo + span * (size_t)t / (size_t)T; *b = lo + span * (size_t)(t + 1) / (size_t)T; };
	auto run = [&](bool write) {
		auto job = [&](int t) {
			size_t a, b; slice(t, &a, &b);
			if (!write) cnt[(size_t)t + 1] = nl_scan(w, a, b, nullptr);
			else (void)nl_scan(w, a, b, s->nl.p + s->nl.n + cnt[(size_t)t]);
		};
		if (T == 1) job(0);
		else par_for(T, job);
	};
	run(false);
	for (int t = 0; t < T; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
	s->nl.reserve(s->nl.n + cnt[(size_t)T]);
	run(true);
	s->nl.n += cnt[(size_t)T];
	s->idx_end = hi;
	g_io_index_s += io_now() - t0;
}
/* the next '\n' at or after window offset p, or (size_t)-1 when the window holds none (what memchr said before) */
static inline size_t nl_next(BtReadStream* s, size_t p)
{
	while (s->nl_cur < s->nl.size() && s->nl[s->nl_cur] < p) s->nl_cur++;
	return s->nl_cur < s->nl.size() ? s->nl[s->nl_cur] : (size_t)-1;
}

/* more of the file into the window (and into the newline index); `want` = bytes the batch still expects to need */
static bool fq_more(BtReadStream* s, size_t want = 0)
{
	if (s->feof || !s->f) return false;
	/* no further ahead than the batch needs (what a batch leaves over is moved to the front of the window by the next one),
	 * in pieces large enough for several threads to share */
	static const size_t fillMin = io_knob("BT_IO_FILL_BYTES", 64u << 20), sliceMin = io_knob("BT_IO_SLICE_BYTES", 2u << 20) * 4u;
	size_t room = want + fillMin / 16u;
	if (room < fillMin) room = fillMin;
	if (room > (1u << 30)) room = 1u << 30;
	if (s->buf.size() - s->end < room) s->buf.resize(s->end + room + (s->buf.size() < (256u << 20) ? (32u << 20) : s->buf.size() / 8u));
	const double t0 = io_now();
	size_t got = 0;
	if (s->rawfd >= 0) {
		/* pread by several threads, a slice each; a short slice is the end of the file */
		int T = s->io_threads < 1 ? 1 : (s->io_threads > 16 ? 16 : s->io_threads);
		if ((size_t)T > room / sliceMin) T = (int)(room / sliceMin);
		if (T < 1) T = 1;
		std::vector<size_t> g((size_t)T, 0);
		char* dst = s->buf.data() + s->end;
		auto rd = [&](int t) {
			const size_t a = room * (size_t)t / (size_t)T, b = room * (size_t)(t + 1) / (size_t)T;
			size_t done = 0;
			while (a + done < b) {
				const ssize_t k = pread(s->rawfd, dst + a + done, b - a - done, (off_t)(s->rawoff + a + done));
				if (k <= 0) break;
				done += (size_t)k;
			}
			g[(size_t)t] = done;
		};
		if (T == 1) rd(0);
		else par_for(T, rd);
		for (int t = 0; t < T; t++) {
			const size_t a = room * (size_t)t / (size_t)T, b = room * (size_t)(t + 1) / (size_t)T;
			got += g[(size_t)t];
			if (g[(size_t)t] < b - a) break;               /* the file ended inside this slice: what lies beyond was not read */
		}
		s->rawoff += got;
	} else {
		const int k = gzread(s->f, s->buf.data() + s->end, (unsigned)(room > (64u << 20) ? (64u << 20) : (fillMin < (64u << 20) ? fillMin : room)));
		got = k > 0 ? (size_t)k : 0;
	}
	g_io_fill_s += io_now() - t0;
	if (got == 0) { s->feof = true; return false; }
	s->end += got;
	nl_index_more(s);
	return true;
}

static int next_fastq(BtReadStream* s, uint32_t max_reads, int threads, BtHostBatch* batch, std::string* err)
{
	static const bool prof = getenv("BT_IO_PROFILE") != nullptr;
	const double tp0 = prof ? io_now() : 0; const double fill0 = g_io_fill_s, index0 = g_io_index_s;
	s->io_threads = threads;
	/* the window keeps only what the previous batch did not use; the newline index moves with it */
	if (s->pos > 0) {
		const size_t by = s->pos;
		memmove(s->buf.data(), s->buf.data() + by, s->end - by); s->end -= by; s->pos = 0;
		size_t k = 0;
		while (s->nl_cur < s->nl.size() && s->nl[s->nl_cur] < by) s->nl_cur++;
		for (size_t i = s->nl_cur; i < s->nl.size(); i++) s->nl[k++] = s->nl[i] - by;
		s->nl.resize(k); s->nl_cur = 0;
		s->idx_end = s->idx_end > by ? s->idx_end - by : 0;
	}
	if (s->idx_end < s->end) nl_index_more(s);        /* (a window another path of the reader filled) */
	BtPod<FqRec> recs;
	recs.reserve(max_reads < (1u << 22) ? max_reads : (1u << 22));
	uint32_t maxline = 1;
	bool at_limit = false;
	static const size_t bulkMin = io_knob("BT_IO_BULK_MIN", 4096);
	while (!s->done && recs.size() < max_reads) {
		if (s->rdid >= s->limit) { s->done = true; at_limit = true; break; }
		if (!s->f) {
			/* the next file continues the same window: this batch's records stay where they are */
			if (!st_open_next(s, err, true)) { s->done = true; break; }
		}
		if (s->file_first) {
			for (;;) {
				while (s->pos < s->end && (s->buf[s->pos] == '\r' || s->buf[s->pos] == '\n')) s->pos++;
				if (s->pos < s->end || !fq_more(s, (size_t)((double)(max_reads - recs.size()) * s->avg_rec))) break;
			}
			if (s->pos >= s->end || s->buf[s->pos] != '@') { *err = "Error: reads file does not look like a FASTQ file"; return BT_ERR_READS; }
			s->file_first = false;
		}
		if (s->rdid >= s->o.skip) {
			/* the bulk of a batch: as many whole records as the newline index already holds -- four newlines each, one after
			 * the other from here -- are laid out by the io threads, a share each; whatever needs a look (the end of a file,
			 * more data, -s / -u) goes through the loop below, record by record, as before */
			while (s->nl_cur < s->nl.size() && s->nl[s->nl_cur] < s->pos) s->nl_cur++;
			size_t m = (s->nl.size() - s->nl_cur) / 4u;
			if (m > max_reads - recs.size()) m = max_reads - recs.size();
			if ((uint64_t)m > s->limit - s->rdid) m = (size_t)(s->limit - s->rdid);
			if (m >= bulkMin) {
				const size_t base = recs.size(), c0 = s->nl_cur, p0 = s->pos;
				const uint64_t id0 = s->rdid;
				recs.resize_uninit(base + m);
				FqRec* out = recs.data() + base;
				const size_t* nl = s->nl.p;
				int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
				if ((size_t)T > m / 2048u) T = (int)(m / 2048u);
				if (T < 1) T = 1;
				std::vector<uint32_t> mx((size_t)T, 1u);
				auto job = [&](int t) {
					const size_t lo = m * (size_t)t / (size_t)T, hi = m * (size_t)(t + 1) / (size_t)T;
					uint32_t ml = 1;
					for (size_t i = lo; i < hi; i++) {
						const size_t off = i == 0 ? p0 : nl[c0 + 4u * i - 1u] + 1u;
						FqRec& r = out[i];
						r.off = off; r.rdid = id0 + i;
						for (int k = 0; k < 4; k++) r.e[k] = (uint32_t)(nl[c0 + 4u * i + (size_t)k] - off);
						const uint32_t l2 = r.e[1] - r.e[0] - 1u;
						if (l2 > ml) ml = l2;
					}
					mx[(size_t)t] = ml;
				};
				if (T == 1) job(0);
				else par_for(T, job);
				for (int t = 0; t < T; t++) if (mx[(size_t)t] > maxline) maxline = mx[(size_t)t];
				s->pos = nl[c0 + 4u * m - 1u] + 1u;
				s->nl_cur = c0 + 4u * m;
				s->rdid += m; s->file_recs += m;
				if (s->pos >= s->end && s->feof) st_close_file(s);
				continue;
			}
		}
		FqRec r; r.off = s->pos; r.rdid = s->rdid;
		size_t p = s->pos; int k = 0; bool partial = false;
		while (k < 4) {
			const size_t nl = nl_next(s, p);
			if (nl != (size_t)-1) { r.e[k++] = (uint32_t)(nl - r.off); p = nl + 1; continue; }
			if (fq_more(s, (size_t)((double)(max_reads - recs.size()) * s->avg_rec))) continue;
			/* end of file: it stands in for the fourth newline only */
			if (k == 3) { r.e[3] = (uint32_t)(s->end - r.off); k = 4; p = s->end; }
			else partial = true;
			break;
		}
		if (partial) {
			/* A file that ends inside a record.  The reference's light parser (pat.cpp:822-858) then
			 * also gives up the record before it, unless that one closed a 16-read batch
			 * (--reads-per-batch); its read id is handed to the next file's first read. */
			if (k > 0 && s->file_recs % 16u != 0) {      /* k == 0: the file ended before the first newline of a record -- a clean end */
				s->rdid--;
				if (s->rdid >= s->o.skip && !recs.empty() && recs.back().rdid == s->rdid) recs.pop_back();
			}
			s->pos = s->end; st_close_file(s); continue;
		}
		s->pos = p;
		if (s->rdid >= s->o.skip) {
			recs.push_back(r);
			const uint32_t l2 = r.e[1] - r.e[0] - 1u;
			if (l2 > maxline) maxline = l2;
		}
		s->rdid++; s->file_recs++;
		if (s->pos >= s->end && s->feof) st_close_file(s);
	}
	if ((at_limit || (recs.size() == max_reads && !s->done)) && s->f && s->file_recs % 16u != 0) {
		/* the batch is full (or -u is reached: the reference's reader is a batch ahead of that): look (without consuming)
		 * whether the file ends inside the next record, because that would take this batch's last record with it */
		size_t p = s->pos; int k = 0;
		const size_t cur0 = s->nl_cur;
		while (k < 3) {
			const size_t nl = nl_next(s, p);
			if (nl != (size_t)-1) { k++; p = nl + 1; continue; }
			if (fq_more(s)) continue;
			break;
		}
		s->nl_cur = cur0;                                  /* a look ahead: nothing consumed */
		if (k > 0 && k < 3 && s->feof) {
			s->rdid--;
			if (!recs.empty() && recs.back().rdid == s->rdid) recs.pop_back();
			s->pos = s->end; st_close_file(s);
		}
	}
	const size_t n = recs.size();
	if (n == 0) {
		batch->n = 0;
		if (s->open_failed) { *err = "Error: could not open the remaining read file(s)"; return BT_ERR_READS; }
		return BT_OK;
	}
	const double tp1 = prof ? io_now() : 0;
	if (s->pos > recs.front().off) s->avg_rec = (double)(s->pos - recs.front().off) / (double)n + 1.0;
	if (maxline > 1040u) maxline = 1040u;
	const uint32_t stride = (maxline + 15u) & ~15u;
	batch->reset((uint32_t)n, stride);
	batch->rdid.resize(n);
	std::vector<uint32_t> name_b(n), name_n(n);
	std::vector<std::string> alt_name;                 /* default names (read id): rare */
	std::vector<size_t> alt_at;
	const int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
	std::vector<std::string> errs((size_t)T);
	std::vector<size_t> err_at((size_t)T, (size_t)-1);
	const uint8_t* a2d = asc2dna_table();
	uint8_t seqcode[256];
	for (int c = 0; c < 256; c++) seqcode[c] = (isalpha(c) || c == '.') ? a2d[c == '.' ? 'N' : c] : 0xff;
	const bt_read_opts o = s->o;
	const char* W = s->buf.data();
	const double tp2 = prof ? io_now() : 0;
	auto work = [&](int t) {
		const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
		for (size_t i = lo; i < hi; i++) {
			const FqRec& r = recs[i];
			const char* rec = W + r.off;
			const uint32_t len = r.e[3] + 1u > (uint32_t)(s->end - r.off) ? r.e[3] : r.e[3] + 1u;   /* with its last '\n' if present */
			uint8_t* sq = batch->seq + i * stride;
			uint8_t* ql = batch->qual + i * stride;
			const uint32_t L = r.e[1] - r.e[0] - 1u, LQ = r.e[3] - r.e[2] - 1u;
			const uint8_t* sl = (const uint8_t*)rec + r.e[0] + 1;
			const uint8_t* qlin = (const uint8_t*)rec + r.e[2] + 1;
			bool fast = r.e[0] > 1 && L == LQ && L <= 1024u && rec[r.e[1] + 1] == '+' && memchr(rec, '\r', r.e[3]) == nullptr;
			uint32_t out_len = 0;
			if (fast) {
				const uint32_t t5 = (uint32_t)o.trim5 < L ? (uint32_t)o.trim5 : L;
				const uint32_t kept = L - t5;
				const uint32_t t3 = (uint32_t)o.trim3 < kept ? (uint32_t)o.trim3 : kept;
				out_len = kept - t3;
				uint8_t bad = 0;
				for (uint32_t j = 0; j < L; j++) bad |= seqcode[sl[j]];          /* 0xff marks a character the parser drops */
				if (bad == 0xff) fast = false;
				else {
					for (uint32_t j = 0; j < out_len; j++) sq[j] = seqcode[sl[t5 + j]];
					if (o.qual_enc == BT_QUAL_PHRED33) {
						uint8_t mn = 255;
						for (uint32_t j = 0; j < L; j++) mn = qlin[j] < mn ? qlin[j] : mn;
						if (L && mn < 33) fast = false;
						else memcpy(ql, qlin + t5, out_len);
					} else if (o.qual_enc == BT_QUAL_PHRED64) {
						uint8_t mn = 255;
						for (uint32_t j = 0; j < L; j++) mn = qlin[j] < mn ? qlin[j] : mn;
						if (L && mn < 64) fast = false;
						else for (uint32_t j = 0; j < out_len; j++) ql[j] = (uint8_t)(qlin[t5 + j] - 31);
					} else fast = false;
					if (L == 0) fast = false;           /* an empty read has its own story in the reference parser */
				}
			}
			if (fast) {
				name_b[i] = 1; name_n[i] = r.e[0] - 1u;
			} else {
				BtParsed p;
				if (!parse_fastq(rec, len, o, r.rdid, &p, &errs[(size_t)t])) { err_at[(size_t)t] = i; return; }
				if (p.seq.size() > 1024 || p.seq.size() > stride) {
					errs[(size_t)t] = "Reads file contained a pattern with more than 1024 sequence characters.\n"
					                  "Please truncate reads and quality values and and re-run Bowtie.\n"
					                  "Offending read: " + name_of(rec, p, r.rdid);
					err_at[(size_t)t] = i; return;
				}
				out_len = (uint32_t)p.seq.size();
				memcpy(sq, p.seq.data(), out_len); memcpy(ql, p.qual.data(), out_len);
				name_b[i] = (uint32_t)p.name_b; name_n[i] = (uint32_t)p.name_n;
			}
			memset(sq + out_len, 4, stride - out_len);
			memset(ql + out_len, 33, stride - out_len);
			batch->len[i] = (uint16_t)out_len;
			batch->rdid[i] = r.rdid;
			if (name_n[i]) batch->seed[i] = rand_seed(sq, ql, out_len, rec + name_b[i], name_n[i], o.seed);
		}
	};
	if (T == 1 || n < 4096) { for (int t = 0; t < T; t++) { work(t); if (err_at[(size_t)t] != (size_t)-1) break; } }
	else par_for(T, work);
	size_t first_err = (size_t)-1; int et = -1;
	for (int t = 0; t < T; t++) if (err_at[(size_t)t] < first_err) { first_err = err_at[(size_t)t]; et = t; }
	if (et >= 0) { *err = errs[(size_t)et]; batch->n = 0; return BT_ERR_READS; }
	const double tp3 = prof ? io_now() : 0;
	/* names: the offsets are a running sum (sequential, short); the copies are not */
	batch->name_off.resize(n + 1);
	uint64_t tot = 0;
	for (size_t i = 0; i < n; i++) {
		batch->name_off[i] = tot;
		if (name_n[i]) tot += name_n[i];
		else { char b[24]; tot += (uint64_t)snprintf(b, sizeof(b), "%llu", (unsigned long long)recs[i].rdid); }
	}
	batch->name_off[n] = tot;
	batch->names.resize((size_t)tot);
	auto copy_names = [&](int t) {
		const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
		for (size_t i = lo; i < hi; i++) {
			char* dst = &batch->names[(size_t)batch->name_off[i]];
			if (name_n[i]) memcpy(dst, W + recs[i].off + name_b[i], name_n[i]);
			else {
				char b[24]; const int k = snprintf(b, sizeof(b), "%llu", (unsigned long long)recs[i].rdid);
				memcpy(dst, b, (size_t)k);
				batch->seed[i] = rand_seed(batch->seq + i * stride, batch->qual + i * stride, batch->len[i], b, (size_t)k, o.seed);
			}
		}
	};
	if (T == 1 || n < 4096) { for (int t = 0; t < T; t++) copy_names(t); }
	else par_for(T, copy_names);
	batch->raw_off.clear(); batch->raw.clear();
	if (o.flags & BT_READ_KEEP_RAW) {
		/* the records themselves, for --al/--un/--max: what the reference keeps as readOrigBuf (a final
		 * record that the end of the file closed gets its newline) */
		batch->raw_off.resize(n + 1);
		uint64_t rt = 0;
		for (size_t i = 0; i < n; i++) { batch->raw_off[i] = rt; rt += recs[i].e[3] + 1u; }
		batch->raw_off[n] = rt;
		batch->raw.resize((size_t)rt);
		for (size_t i = 0; i < n; i++) {
			const size_t have = recs[i].off + recs[i].e[3] < s->end ? recs[i].e[3] + 1u : recs[i].e[3];
			memcpy(&batch->raw[(size_t)batch->raw_off[i]], W + recs[i].off, have);
			if (have == recs[i].e[3]) batch->raw[(size_t)batch->raw_off[i] + have] = '\n';
		}
	}
	batch->first_rdid = batch->rdid[0];
	if (prof) fprintf(stderr, "[io] fastq batch of %zu: window+scan %.3f s (of which reading the file %.3f, indexing its newlines %.3f), buffers %.3f, records on %d threads %.3f, names %.3f\n",
	                  n, tp1 - tp0, g_io_fill_s - fill0, g_io_index_s - index0, tp2 - tp1, T, tp3 - tp2, io_now() - tp3);
	return BT_OK;
}


/* Read::fixMateName (read.h:141-165) + finalizePair (pat.cpp:76-88): a mate's name ends in /1 (/2) --
 * appended unless already there -- and its seed is computed over that name */
static void fix_mate_names(BtHostBatch* batch, int mate, uint32_t global_seed)
{
	const uint32_t n = batch->n;
	std::string names;
	std::vector<uint64_t> off((size_t)n + 1);
	names.reserve(batch->names.size() + 2ull * n);
	const char digit = mate == 1 ? '1' : '2';
	for (uint32_t i = 0; i < n; i++) {
		const char* nm = batch->names.data() + batch->name_off[i];
		const size_t nn = (size_t)(batch->name_off[i + 1] - batch->name_off[i]);
		off[i] = names.size();
		names.append(nm, nn);
		/* an unpaired record of a --12 file stays an unpaired read: finalize(), not finalizePair() (pat.cpp:64-88) */
		if (!batch->paired.empty() && !batch->paired[i]) continue;
		if (nn < 2 || nm[nn - 2] != '/' || nm[nn - 1] != digit) { names.push_back('/'); names.push_back(digit); }
	}
	off[n] = names.size();
	batch->names.swap(names);
	batch->name_off.swap(off);
	for (uint32_t i = 0; i < n; i++)
		if (batch->paired.empty() || batch->paired[i]) batch->seed[i] = rand_seed(batch->seq + (size_t)i * batch->stride, batch->qual + (size_t)i * batch->stride, batch->len[i],
		                           batch->names.data() + batch->name_off[i], (size_t)(batch->name_off[i + 1] - batch->name_off[i]), global_seed);
}

static int io_next_impl(BtReadStream* s, uint32_t max_reads, int threads, BtHostBatch* batch, std::string* err);
int bt_io_next(BtReadStream* s, uint32_t max_reads, int threads, BtHostBatch* batch, std::string* err)
{
	const int rc = io_next_impl(s, max_reads, threads, batch, err);
	batch->end_rdid = s->rdid;                    /* also for an empty batch (end of input) */
	if (rc == BT_OK && batch->n > 0 && (s->o.flags & (BT_READ_MATE1 | BT_READ_MATE2)))
		fix_mate_names(batch, (s->o.flags & BT_READ_MATE1) ? 1 : 2, s->o.seed);
	return rc;
}

static int io_next_impl(BtReadStream* s, uint32_t max_reads, int threads, BtHostBatch* batch, std::string* err)
{
	s->raw.clear(); s->recs.clear();
	batch->n = 0; batch->n_paired = 0;
	if (s->done && s->open_failed) { *err = "Error: could not open the remaining read file(s)"; return BT_ERR_READS; }
	if (s->o.format == BT_FMT_FASTQ && !(s->o.flags & (BT_READ_CAREFUL | BT_READ_INTERLEAVED))) return next_fastq(s, max_reads, threads, batch, err);
	/* ---- light parse (sequential) ---- */
	while (!s->done && s->recs.size() < max_reads) {
		if (s->rdid >= s->limit) {
			s->done = true;
			/* -u reached: the reference's FASTQ reader is up to a batch ahead, and a file that ends inside the next
			 * record takes the last read with it there too */
			if (s->o.format == BT_FMT_FASTQ && s->f && s->file_recs % 16u != 0) {
				const size_t off = s->raw.size();
				const int rc = light_fastq(s, err);
				s->raw.resize(off);
				if (rc == -2) { s->rdid--; if (!s->recs.empty() && s->recs.back().rdid == s->rdid) { s->raw.resize(s->recs.back().off); s->recs.pop_back(); } }
			}
			break;
		}
		if (s->o.format == BT_FMT_CMDLINE) {
			if (s->item >= s->items.size()) { s->done = true; break; }
			const std::string& it = s->items[s->item++];
			const size_t off = s->raw.size();
			s->raw.append(it);
			if (s->rdid >= s->o.skip) s->recs.push_back({off, (uint32_t)it.size(), s->rdid});
			else s->raw.resize(off);
			s->rdid++;
			continue;
		}
		if (!s->f && !st_open_next(s, err)) { s->done = true; break; }
		const size_t off = s->raw.size();
		int rc;
		if (s->o.format == BT_FMT_FASTQ && (s->o.flags & BT_READ_INTERLEAVED)) {
			/* two records per read id: the first mate's, then the second mate's (pat.cpp:841-851); this stream
			 * keeps the one it was opened for.  A last record without a partner is dropped. */
			rc = light_fastq(s, err);
			if (rc == 1) {
				const size_t mid = s->raw.size();
				const int rc2 = light_fastq(s, err);
				if (rc2 == -1) return BT_ERR_READS;
				if (rc2 <= 0) { s->raw.resize(off); rc = rc2; }
				else if (s->o.flags & BT_READ_MATE2) s->raw.erase(off, mid - off);
				else s->raw.resize(mid);
			}
		}
		else if (s->o.format == BT_FMT_FASTQ) rc = light_fastq(s, err);
		else if (s->o.format == BT_FMT_FASTA) rc = light_fasta(s, err);
		else if (s->o.format == BT_FMT_FASTA_CONT) rc = light_fasta_cont(s, err);
		else rc = light_raw(s, err);
		if (rc == -1) return BT_ERR_READS;
		if (rc == -2) {
			/* the file ended inside a record: the reference's reader drops the read (pair) before it too -- unless that
			 * one closed a 16-read batch -- and hands its read id on (the same rule as in next_fastq) */
			if (s->file_recs % 16u != 0) {
				s->rdid--;
				if (!s->recs.empty() && s->recs.back().rdid == s->rdid) { s->raw.resize(s->recs.back().off); s->recs.pop_back(); }
			}
			rc = 0;
		}
		if (rc == 0) { st_close_file(s); continue; }
		if (s->rdid >= s->o.skip) s->recs.push_back({off, (uint32_t)(s->raw.size() - off), s->rdid});
		else s->raw.resize(off);                    /* skipped reads are never parsed */
		s->rdid++; s->file_recs++;
	}
	const size_t nrec = s->recs.size();
	if (nrec == 0) {
		if (s->open_failed) { *err = "Error: could not open the remaining read file(s)"; return BT_ERR_READS; }
		return BT_OK;
	}

	/* ---- per-record parse (parallel) ---- */
	std::vector<BtParsed> parsed(nrec);
	const int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
	std::vector<std::string> errs((size_t)T);
	std::vector<size_t> err_at((size_t)T, (size_t)-1);
	auto work = [&](int t) {
		const size_t lo = nrec * (size_t)t / (size_t)T, hi = nrec * (size_t)(t + 1) / (size_t)T;
		for (size_t i = lo; i < hi; i++) {
			const BtRec& rc = s->recs[i];
			const char* r = s->raw.data() + rc.off;
			bool ok = true;
			switch (s->o.format) {
			case BT_FMT_FASTQ: ok = parse_fastq(r, rc.len, s->o, rc.rdid, &parsed[i], &errs[(size_t)t]); break;
			case BT_FMT_FASTA: ok = parse_fasta(r, rc.len, s->o, &parsed[i]); break;
			case BT_FMT_RAW: ok = parse_raw(r, rc.len, s->o, &parsed[i]); break;
			case BT_FMT_FASTA_CONT: ok = parse_fasta_cont(r, rc.len, s->o, &parsed[i]); break;
			case BT_FMT_TABBED: ok = parse_tabbed(r, rc.len, s->o, rc.rdid, &parsed[i], &errs[(size_t)t]); break;
			default: ok = parse_cmdline(r, rc.len, s->o, rc.rdid, &parsed[i], &errs[(size_t)t]); break;
			}
			if (ok && parsed[i].seq.size() > 1024) {
				errs[(size_t)t] = "Reads file contained a pattern with more than 1024 sequence characters.\n"
				                  "Please truncate reads and quality values and and re-run Bowtie.\n"
				                  "Offending read: " + name_of(r, parsed[i], rc.rdid);
				ok = false;
			}
			if (!ok) { err_at[(size_t)t] = i; return; }
		}
	};
	if (T == 1 || nrec < 4096) { for (int t = 0; t < T; t++) work(t); }
	else par_for(T, work);
	size_t first_err = (size_t)-1; int et = -1;
	for (int t = 0; t < T; t++) if (err_at[(size_t)t] < first_err) { first_err = err_at[(size_t)t]; et = t; }
	if (et >= 0) { *err = errs[(size_t)et]; return BT_ERR_READS; }

	/* ---- pack ---- */
	std::vector<uint32_t> keep; keep.reserve(nrec);
	size_t maxlen = 1;
	for (size_t i = 0; i < nrec; i++) {
		if (!parsed[i].ok) continue;
		keep.push_back((uint32_t)i);
		if (parsed[i].seq.size() > maxlen) maxlen = parsed[i].seq.size();
	}
	const uint32_t n = (uint32_t)keep.size();
	const uint32_t stride = (uint32_t)((maxlen + 15u) & ~(size_t)15u);
	batch->reset(n, stride);
	batch->n_paired = 0;
	batch->paired.clear();
	if (s->o.format == BT_FMT_TABBED) batch->paired.resize(n);
	for (uint32_t k = 0; k < n; k++) if (parsed[keep[k]].paired) { batch->n_paired++; batch->paired[k] = 1; }
	batch->rdid.resize(n);
	batch->name_off.assign((size_t)n + 1, 0);
	batch->names.clear();
	for (uint32_t k = 0; k < n; k++) {
		const BtParsed& p = parsed[keep[k]];
		const BtRec& rc = s->recs[keep[k]];
		batch->name_off[k] = batch->names.size();
		if (p.name_n || p.name_as_is) batch->names.append(s->raw.data() + rc.off + p.name_b, p.name_n);
		else { char b[24]; snprintf(b, sizeof(b), "%llu", (unsigned long long)rc.rdid); batch->names.append(b); }
		batch->rdid[k] = rc.rdid;
	}
	batch->name_off[n] = batch->names.size();
	batch->raw_off.clear(); batch->raw.clear();
	if (s->o.flags & BT_READ_KEEP_RAW) {
		batch->raw_off.resize((size_t)n + 1);
		for (uint32_t k = 0; k < n; k++) {
			const BtRec& rc = s->recs[keep[k]];
			batch->raw_off[k] = batch->raw.size();
			if (s->o.format == BT_FMT_CMDLINE) {
				/* VectorPatternSource keeps "<id>\t<seq>\t<quals>" (pat.cpp:372-392) */
				char b[24]; snprintf(b, sizeof(b), "%llu", (unsigned long long)rc.rdid);
				const char* r = s->raw.data() + rc.off;
				const char* colon = (const char*)memchr(r, ':', rc.len);
				const size_t sl = colon ? (size_t)(colon - r) : rc.len;
				batch->raw.append(b); batch->raw.push_back('\t'); batch->raw.append(r, sl); batch->raw.push_back('\t');
				if (colon) batch->raw.append(colon + 1, rc.len - sl - 1); else batch->raw.append(sl, 'I');
			} else batch->raw.append(s->raw.data() + rc.off, rc.len);
		}
		batch->raw_off[n] = batch->raw.size();
	}
	auto pack = [&](int t) {
		const size_t lo = (size_t)n * (size_t)t / (size_t)T, hi = (size_t)n * (size_t)(t + 1) / (size_t)T;
		for (size_t k = lo; k < hi; k++) {
			const BtParsed& p = parsed[keep[k]];
			uint8_t* sq = batch->seq + k * stride;
			uint8_t* ql = batch->qual + k * stride;
			const size_t L = p.seq.size();
			memcpy(sq, p.seq.data(), L); memset(sq + L, 4, stride - L);
			memcpy(ql, p.qual.data(), L); memset(ql + L, 33, stride - L);
			batch->len[k] = (uint16_t)L;
			batch->seed[k] = rand_seed(sq, ql, L, batch->names.data() + batch->name_off[k],
			                           (size_t)(batch->name_off[k + 1] - batch->name_off[k]), s->o.seed);
		}
	};
	if (T == 1 || n < 4096) { for (int t = 0; t < T; t++) pack(t); }
	else par_for(T, pack);
	batch->first_rdid = n ? batch->rdid[0] : 0;
	return BT_OK;
}

/* ---- output ---------------------------------------------------------------------------------- */
static inline void put_u(std::string* o, uint64_t v)
{
	if (v < 10u) { o->push_back((char)('0' + v)); return; }
	char b[24]; int n = 24;
	do { b[--n] = (char)('0' + v % 10u); v /= 10u; } while (v);
	o->append(b + n, (size_t)(24 - n));
}
static inline void put_i(std::string* o, int64_t v)
{
	if (v < 0) { o->push_back('-'); put_u(o, (uint64_t)(-v)); } else put_u(o, (uint64_t)v);
}
static inline void put_ref(std::string* o, const BtRefNames& refs, uint32_t tidx, const bt_out_opts& op)
{
	if (!op.ref_idx && tidx < refs.names.size()) {
		const std::string& nm = refs.names[tidx];
		if (op.full_ref) o->append(nm);
		else { size_t i = 0; while (i < nm.size() && !isspace((unsigned char)nm[i])) i++; o->append(nm, 0, i); }
	} else put_u(o, tidx);
}
static inline void put_qname(std::string* o, const char* nm, size_t n, bool trunc)
{
	size_t i = 0;
	if (trunc) { while (i < n && !isspace((unsigned char)nm[i])) i++; } else i = n;
	o->append(nm, i);
}
/* the read as aligned: reverse-complemented / reversed for '-' hits (Hit::patSeq, Hit::quals) */
static inline void put_seq(std::string* o, const uint8_t* seq, uint32_t L, bool fw)
{
	/* one resize, then plain stores: a push_back per character is most of a formatter thread's time */
	static const char fwc[] = "ACGTN", rcc[] = "TGCAN";
	const size_t at = o->size();
	o->resize(at + L);
	char* d = &(*o)[0] + at;
	if (fw) for (uint32_t i = 0; i < L; i++) d[i] = fwc[seq[i] > 4 ? 4 : seq[i]];
	else for (uint32_t i = 0; i < L; i++) d[i] = rcc[seq[L - 1u - i] > 4 ? 4 : seq[L - 1u - i]];
}
static inline void put_qual(std::string* o, const uint8_t* q, uint32_t L, bool fw)
{
	if (fw) { o->append((const char*)q, L); return; }
	const size_t at = o->size();
	o->resize(at + L);
	char* d = &(*o)[0] + at;
	for (uint32_t i = 0; i < L; i++) d[i] = (char)q[L - 1u - i];
}

struct MmList { uint32_t n; uint16_t e[64]; };

static void verbose_hit(std::string* o, const char* nm, size_t nn, const uint8_t* seq, const uint8_t* qual, uint32_t L,
                        const bt_hit& h, const uint16_t* mm, uint32_t seed, const BtRefNames& refs, const bt_out_opts& op)
{
	uint32_t field = 0; bool first = true;
#define FIELD(body) do { if (!((op.suppress >> field++) & 1ull)) { if (first) first = false; else o->push_back('\t'); body; } } while (0)
	const bool fw = h.fw != 0;
	FIELD(o->append(nm, nn));
	FIELD(o->push_back(fw ? '+' : '-'));
	FIELD(put_ref(o, refs, h.tidx, op));
	FIELD(put_u(o, (uint64_t)((int64_t)h.toff + op.off_base)));
	FIELD(put_seq(o, seq, L, fw));
	FIELD(put_qual(o, qual, L, fw));
	FIELD(put_u(o, h.oms));
	auto put_mms = [&]() {
		/* mismatches by ascending 5'-relative offset: pos:ref>read */
		static const char dna[] = "ACGTN", rc[] = "TGCAN";
		bool firstmm = true;
		const uint16_t* sorted = mm;               /* the search stores the list ordered by position */
		const uint32_t n = h.nmm;
		for (uint32_t i = 0; i < n; i++) {
			const uint32_t pos = BT_MM_POS(sorted[i]), refc = BT_MM_REFC(sorted[i]);
			if (!firstmm) o->push_back(',');
			put_u(o, pos);
			o->push_back(':'); o->push_back(dna[refc]); o->push_back('>');
			/* the read character as printed in the sequence column */
			const uint8_t c = seq[pos] > 4 ? 4 : seq[pos];
			o->push_back(fw ? dna[c] : rc[c]);
			firstmm = false;
		}
	};
	FIELD(put_mms());
	if (op.print_cost) {
		FIELD(put_u(o, h.stratum));
		FIELD(put_u(o, h.cost));
	}
	if (op.show_seed) FIELD(put_u(o, seed));
#undef FIELD
	o->push_back('\n');
}

/* The two records an unpaired SAM run consists of are written with plain stores into room made once per record (its size is
 * bounded by the name, the two L-character columns and the reference's name): ~30 appends per record, each with its own
 * capacity check, were what a formatter thread spent its time on (round 6: 3 us per record and thread on the GPU host). */
static inline char* w_u(char* p, uint64_t v)
{
	if (v < 10u) { *p++ = (char)('0' + v); return p; }
	char b[24]; int n = 24;
	do { b[--n] = (char)('0' + v % 10u); v /= 10u; } while (v);
	memcpy(p, b + n, (size_t)(24 - n));
	return p + (24 - n);
}
static inline char* w_s(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }
#define W_LIT(p, lit) w_s((p), (lit), sizeof(lit) - 1u)
static inline char* w_seq(char* d, const uint8_t* seq, uint32_t L, bool fw)
{
	static const char fwc[] = "ACGTN", rcc[] = "TGCAN";
	if (fw) for (uint32_t i = 0; i < L; i++) d[i] = fwc[seq[i] > 4 ? 4 : seq[i]];
	else for (uint32_t i = 0; i < L; i++) d[i] = rcc[seq[L - 1u - i] > 4 ? 4 : seq[L - 1u - i]];
	return d + L;
}
static inline char* w_qual(char* d, const uint8_t* q, uint32_t L, bool fw)
{
	if (fw) { memcpy(d, q, L); return d + L; }
	for (uint32_t i = 0; i < L; i++) d[i] = (char)q[L - 1u - i];
	return d + L;
}
static inline size_t ref_name_len(const BtRefNames& refs, uint32_t tidx, const bt_out_opts& op)
{
	return (!op.ref_idx && tidx < refs.names.size()) ? refs.names[tidx].size() : 12u;
}
static inline char* w_ref(char* p, const BtRefNames& refs, uint32_t tidx, const bt_out_opts& op)
{
	if (!op.ref_idx && tidx < refs.names.size()) {
		const std::string& nm = refs.names[tidx];
		size_t i = nm.size();
		if (!op.full_ref) { i = 0; while (i < nm.size() && !isspace((unsigned char)nm[i])) i++; }
		return w_s(p, nm.data(), i);
	}
	return w_u(p, tidx);
}
static inline char* w_qname(char* p, const char* nm, size_t n, bool trunc)
{
	size_t i = 0;
	if (trunc) { while (i < n && !isspace((unsigned char)nm[i])) i++; } else i = n;
	return w_s(p, nm, i);
}

static void sam_hit(std::string* o, const char* nm, size_t nn, const uint8_t* seq, const uint8_t* qual, uint32_t L,
                    const bt_hit& h, const uint16_t* mm, uint32_t xms, const BtRefNames& refs, const bt_out_opts& op,
                    int mapq_override = -1)
{
	static const char dna[] = "ACGT";
	const bool fw = h.fw != 0;
	const uint32_t n = h.nmm;
	const size_t at = o->size();
	o->resize(at + nn + 2u * (size_t)L + ref_name_len(refs, h.tidx, op) + 6u * (size_t)n + 200u);
	char* const base = &(*o)[0] + at;
	char* p = base;
	p = w_qname(p, nm, nn, !op.no_qname_trunc);
	*p++ = '\t'; p = w_u(p, fw ? 0u : 16u);
	*p++ = '\t'; p = w_ref(p, refs, h.tidx, op);
	*p++ = '\t'; p = w_u(p, (uint64_t)h.toff + 1u);
	*p++ = '\t';
	{ const int64_t mq = mapq_override >= 0 ? mapq_override : op.mapq; if (mq < 0) { *p++ = '-'; p = w_u(p, (uint64_t)(-mq)); } else p = w_u(p, (uint64_t)mq); }
	*p++ = '\t'; p = w_u(p, L); p = W_LIT(p, "M\t*\t0\t0\t");
	p = w_seq(p, seq, L, fw);
	*p++ = '\t';
	p = w_qual(p, qual, L, fw);
	p = W_LIT(p, "\tXA:i:"); p = w_u(p, h.stratum);
	p = W_LIT(p, "\tMD:Z:");
	/* MD walks the alignment left to right on the reference: by 5' offset for '+', by descending
	 * offset for '-' */
	const uint16_t* sorted = mm;                   /* ordered by position (any length: Phred<5 mismatches cost nothing) */
	uint32_t run_from = 0;     /* alignment columns consumed so far */
	for (uint32_t k = 0; k < n; k++) {
		const uint16_t e = fw ? sorted[k] : sorted[n - 1 - k];
		const uint32_t col = fw ? BT_MM_POS(e) : (L - 1u - BT_MM_POS(e));
		p = w_u(p, col - run_from);
		*p++ = dna[BT_MM_REFC(e)];
		run_from = col + 1u;
	}
	p = w_u(p, L - run_from);
	p = W_LIT(p, "\tNM:i:"); p = w_u(p, n);
	if (xms > 0) { p = W_LIT(p, "\tXM:i:"); p = w_u(p, xms); }
	*p++ = '\n';
	o->resize(at + (size_t)(p - base));
}

static void sam_unaligned(std::string* o, const char* nm, size_t nn, const uint8_t* seq, const uint8_t* qual, uint32_t L,
                          uint32_t xm, const bt_out_opts& op)
{
	const size_t at = o->size();
	o->resize(at + nn + 2u * (size_t)L + 64u);
	char* const base = &(*o)[0] + at;
	char* p = base;
	p = w_qname(p, nm, nn, !op.no_qname_trunc);
	p = W_LIT(p, "\t4\t*\t0\t0\t*\t*\t0\t0\t");
	p = w_seq(p, seq, L, true);
	*p++ = '\t';
	p = w_qual(p, qual, L, true);
	p = W_LIT(p, "\tXM:i:"); p = w_u(p, xm);
	*p++ = '\n';
	o->resize(at + (size_t)(p - base));
}

void bt_io_format(const bt_read_batch& rb, const char* names, const uint64_t* name_off, const bt_hit_batch& hb,
                  const BtRefNames& refs, const bt_out_opts& op, uint32_t lo, uint32_t hi, std::string* out,
                  bt_out_tally* tally)
{
	const uint32_t lim = op.all_hits ? hb.hit_cap : (hb.hit_cap < op.khits ? hb.hit_cap : op.khits);
	for (uint32_t i = lo; i < hi; i++) {
		const uint32_t tot = hb.n_hits[i];
		const uint8_t* seq = rb.seq + (size_t)i * rb.stride;
		const uint8_t* qual = rb.qual + (size_t)i * rb.stride;
		const uint32_t L = rb.len[i];
		const char* nm = names + name_off[i];
		const size_t nn = (size_t)(name_off[i + 1] - name_off[i]);
		if (tot == 0) {
			if (tally) tally->unaligned++;
			if (op.sam && !op.no_unal) sam_unaligned(out, nm, nn, seq, qual, L, 0, op);
			continue;
		}
		if (tot > op.mhits) {           /* over the -m ceiling: counted, nothing printed (hit.h:494-500) */
			if (tally) tally->maxed++;
			if (op.sample_max) {
				/* -M (VerboseHitSink::reportMaxed hit.cpp:16-68, SAMHitSink::reportMaxed sam.cpp:263-311): the
				 * first mhits hits were buffered, best stratum first; one of those tied for the best
				 * stratum is printed, picked with the first draw of the read's generator */
				const uint32_t nb = op.mhits < hb.hit_cap ? op.mhits : hb.hit_cap;
				const bt_hit* hs = hb.hits + (size_t)i * hb.hit_cap;
				uint32_t num = 1;
				for (uint32_t k = 1; k < nb; k++) { if (hs[k].stratum == hs[k - 1].stratum) num++; else break; }
				uint32_t last = rb.seed[i];
				last = 1664525u * last + 1013904223u;
				uint32_t r = last >> 16;
				last = 1664525u * last + 1013904223u;
				r = (r ^ last) % num;
				bt_hit h = hs[r];
				const uint16_t* mm = hb.mm_pool ? hb.mm_pool + h.mm_off : nullptr;
				if (op.sam) sam_hit(out, nm, nn, seq, qual, L, h, mm, nb + 1u, refs, op, 0);
				else { h.oms = nb; verbose_hit(out, nm, nn, seq, qual, L, h, mm, rb.seed[i], refs, op); }
				if (tally) { tally->aligned++; tally->reported++; tally->sample_max = 1; }
			}
			continue;
		}
		const uint32_t np = tot < lim ? tot : lim;
		if (tally) { tally->aligned++; tally->reported += np; }
		for (uint32_t k = 0; k < np; k++) {
			const bt_hit& h = hb.hits[(size_t)i * hb.hit_cap + k];
			const uint16_t* mm = hb.mm_pool ? hb.mm_pool + h.mm_off : nullptr;
			if (op.sam) sam_hit(out, nm, nn, seq, qual, L, h, mm, np, refs, op);
			else verbose_hit(out, nm, nn, seq, qual, L, h, mm, rb.seed[i], refs, op);
		}
	}
}

/* SAMHitSink::append for a mate of a paired alignment (sam.cpp:129-257): QNAME without its /1 or /2,
 * FLAG 1|2|64/128 (+16, +32), MRNM '=', MPOS, ISIZE */
static void sam_pair_hit(std::string* o, const char* nm, size_t nn, const uint8_t* seq, const uint8_t* qual, uint32_t L,
                         const bt_hit& h, const uint16_t* mm, const bt_hit& mh, uint32_t mlen, uint32_t xms,
                         const BtRefNames& refs, const bt_out_opts& op, int mapq_override = -1)
{
	static const char dna[] = "ACGT";
	const bool fw = h.fw != 0;
	put_qname(o, nm, nn >= 2 ? nn - 2 : 0, !op.no_qname_trunc);
	const uint32_t flags = 1u | 2u | (h.pad[0] == 1 ? 64u : 128u) | (fw ? 0u : 16u) | (mh.fw ? 0u : 32u);
	o->push_back('\t'); put_u(o, flags);
	o->push_back('\t'); put_ref(o, refs, h.tidx, op);
	o->push_back('\t'); put_u(o, (uint64_t)h.toff + 1u);
	o->push_back('\t'); put_i(o, mapq_override >= 0 ? mapq_override : op.mapq);
	o->push_back('\t'); put_u(o, L); o->append("M\t=\t");
	put_u(o, (uint64_t)mh.toff + 1u);
	o->push_back('\t');
	{
		int64_t ins;
		if (h.toff > mh.toff) ins = -((int64_t)h.toff - (int64_t)mh.toff + (int64_t)L);
		else ins = (int64_t)mh.toff - (int64_t)h.toff + (int64_t)mlen;
		put_i(o, (int64_t)ins);
	}
	o->push_back('\t');
	put_seq(o, seq, L, fw);
	o->push_back('\t');
	put_qual(o, qual, L, fw);
	o->append("\tXA:i:"); put_u(o, h.stratum);
	o->append("\tMD:Z:");
	const uint32_t n = h.nmm;
	uint32_t run_from = 0;
	for (uint32_t k = 0; k < n; k++) {
		const uint16_t e = fw ? mm[k] : mm[n - 1 - k];
		const uint32_t col = fw ? BT_MM_POS(e) : (L - 1u - BT_MM_POS(e));
		put_u(o, col - run_from);
		o->push_back(dna[BT_MM_REFC(e)]);
		run_from = col + 1u;
	}
	put_u(o, L - run_from);
	o->append("\tNM:i:"); put_u(o, n);
	if (xms > 0) { o->append("\tXM:i:"); put_u(o, xms); }
	o->push_back('\n');
}

/* one mate of a pair that did not align (SAMHitSink::reportUnOrMax, sam.cpp:57-124): FLAG 77 / 141 */
static void sam_pair_unaligned(std::string* o, const char* nm, size_t nn, const uint8_t* seq, const uint8_t* qual, uint32_t L,
                               int mate, const bt_out_opts& op)
{
	put_qname(o, nm, nn >= 2 ? nn - 2 : 0, !op.no_qname_trunc);
	o->append(mate == 1 ? "\t77\t*\t0\t0\t*\t*\t0\t0\t" : "\t141\t*\t0\t0\t*\t*\t0\t0\t");
	put_seq(o, seq, L, true);
	o->push_back('\t');
	put_qual(o, qual, L, true);
	o->append("\tXM:i:0\n");
}

void bt_io_format_pairs(const bt_read_batch& r1, const char* names1, const uint64_t* off1,
                        const bt_read_batch& r2, const char* names2, const uint64_t* off2, const bt_hit_batch& hb,
                        const BtRefNames& refs, const bt_out_opts& op, uint32_t lo, uint32_t hi, std::string* out,
                        bt_out_tally* tally)
{
	/* finishRead with createMult(2)'s doubled limits (hit.h:741-786, 1012-1016) */
	const uint32_t maxv = op.mhits == 0xffffffffu ? 0xffffffffu : op.mhits * 2u;
	const uint32_t lim = op.all_hits ? hb.hit_cap : (hb.hit_cap < op.khits * 2u ? hb.hit_cap : op.khits * 2u);
	for (uint32_t i = lo; i < hi; i++) {
		const uint32_t tot = hb.n_hits[i];
		const bt_read_batch* rb[2] = { &r1, &r2 };
		const char* nm[2] = { names1 + off1[i], names2 + off2[i] };
		const size_t nn[2] = { (size_t)(off1[i + 1] - off1[i]), (size_t)(off2[i + 1] - off2[i]) };
		if (tot == 0) {
			if (tally) tally->unaligned++;
			if (op.sam && !op.no_unal)
				for (int m = 0; m < 2; m++)
					sam_pair_unaligned(out, nm[m], nn[m], rb[m]->seq + (size_t)i * rb[m]->stride, rb[m]->qual + (size_t)i * rb[m]->stride,
					                   rb[m]->len[i], m + 1, op);
			continue;
		}
		if (tot > maxv) {
			if (tally) tally->maxed++;
			if (op.sample_max) {
				/* -M for pairs (VerboseHitSink::reportMaxed hit.cpp:27-55, SAMHitSink::reportMaxed sam.cpp:274-299): of the
				 * -M pairs that were buffered, those whose better mate is in the best stratum seen; one of them, picked with
				 * the first draw of the first mate's generator */
				uint32_t nb = maxv < hb.hit_cap ? maxv : hb.hit_cap;
				nb &= ~1u;
				const bt_hit* hs = hb.hits + (size_t)i * hb.hit_cap;
				uint32_t best = 999, num = 0;
				for (uint32_t k = 0; k + 1 < nb; k += 2) {
					const uint32_t st = hs[k].stratum < hs[k + 1].stratum ? hs[k].stratum : hs[k + 1].stratum;
					if (st < best) { best = st; num = 1; } else if (st == best) num++;
				}
				if (num > 0) {
					uint32_t last = r1.seed[i];
					last = 1664525u * last + 1013904223u;
					uint32_t r = last >> 16;
					last = 1664525u * last + 1013904223u;
					r = (r ^ last) % num;
					uint32_t seen = 0;
					for (uint32_t k = 0; k + 1 < nb; k += 2) {
						const uint32_t st = hs[k].stratum < hs[k + 1].stratum ? hs[k].stratum : hs[k + 1].stratum;
						if (st != best) continue;
						if (seen++ != r) continue;
						for (uint32_t e = 0; e < 2; e++) {
							bt_hit h = hs[k + e];
							const bt_hit& mh = hs[k + (e ^ 1u)];
							const int m = h.pad[0] == 2 ? 1 : 0;
							const uint8_t* seq = rb[m]->seq + (size_t)i * rb[m]->stride;
							const uint8_t* qual = rb[m]->qual + (size_t)i * rb[m]->stride;
							const uint16_t* mm = hb.mm_pool ? hb.mm_pool + h.mm_off : nullptr;
							if (op.sam) sam_pair_hit(out, nm[m], nn[m], seq, qual, rb[m]->len[i], h, mm, mh, rb[m ^ 1]->len[i], nb / 2u + 1u, refs, op, 0);
							else { h.oms = nb / 2u; verbose_hit(out, nm[m], nn[m], seq, qual, rb[m]->len[i], h, mm, rb[m]->seed[i], refs, op); }
						}
						break;
					}
					if (tally) { tally->aligned++; tally->reported_paired += 2; tally->sample_max = 1; }
				}
			}
			continue;
		}
		uint32_t np = tot < lim ? tot : lim;
		np &= ~1u;
		if (tally) { tally->aligned++; tally->reported_paired += np; }
		for (uint32_t k = 0; k < np; k++) {
			const bt_hit& h = hb.hits[(size_t)i * hb.hit_cap + k];
			const bt_hit& mh = hb.hits[(size_t)i * hb.hit_cap + (k ^ 1u)];
			const int m = h.pad[0] == 2 ? 1 : 0;
			const uint8_t* seq = rb[m]->seq + (size_t)i * rb[m]->stride;
			const uint8_t* qual = rb[m]->qual + (size_t)i * rb[m]->stride;
			const uint16_t* mm = hb.mm_pool ? hb.mm_pool + h.mm_off : nullptr;
			if (op.sam) sam_pair_hit(out, nm[m], nn[m], seq, qual, rb[m]->len[i], h, mm, mh, rb[m ^ 1]->len[i], np / 2u, refs, op);
			else verbose_hit(out, nm[m], nn[m], seq, qual, rb[m]->len[i], h, mm, rb[m]->seed[i], refs, op);
		}
	}
}

void bt_io_sam_header(const BtRefNames& refs, const bt_out_opts& op, const char* cmdline, const char* rgline,
                      std::string* o)
{
	o->append("@HD\tVN:1.0\tSO:unsorted\n");
	if (!op.sam_nosq) {
		for (size_t i = 0; i < refs.lens.size(); i++) {
			o->append("@SQ\tSN:");
			/* names here even with --refidx, which only changes the alignment records (the header is written from the
			 * names read off the index, ebwt_search.cpp:3220-3226) */
			bt_out_opts named = op; named.ref_idx = 0;
			put_ref(o, refs, (uint32_t)i, named);
			o->append("\tLN:"); put_u(o, refs.lens[i]); o->push_back('\n');
		}
	}
	if (rgline && *rgline) { o->append("@RG\t"); o->append(rgline); o->push_back('\n'); }
	o->append("@PG\tID:Bowtie\tVN:1.3.1\tCL:\""); o->append(cmdline ? cmdline : ""); o->append("\"\n");
}

void bt_io_summary(const bt_out_tally& t, std::string* o)
{
	/* with -M the sampled reads are already among the aligned ones (hit.h:289-319) */
	const bool sm = t.sample_max != 0;
	const uint64_t tot = t.aligned + t.unaligned + (sm ? 0 : t.maxed);
	const uint64_t withAl = t.aligned + (sm ? 0 : t.maxed);
	double al = 0, un = 0, mx = 0;
	if (tot) { al = 100.0 * (double)withAl / (double)tot; un = 100.0 * (double)t.unaligned / (double)tot; mx = 100.0 * (double)t.maxed / (double)tot; }
	char b[256];
	snprintf(b, sizeof(b), "# reads processed: %llu\n", (unsigned long long)tot); o->append(b);
	snprintf(b, sizeof(b), "# reads with at least one alignment: %llu (%.2f%%)\n", (unsigned long long)withAl, al); o->append(b);
	snprintf(b, sizeof(b), "# reads that failed to align: %llu (%.2f%%)\n", (unsigned long long)t.unaligned, un); o->append(b);
	if (t.maxed) {
		snprintf(b, sizeof(b), sm ? "# reads with alignments sampled due to -M: %llu (%.2f%%)\n" : "# reads with alignments suppressed due to -m: %llu (%.2f%%)\n",
		         (unsigned long long)t.maxed, mx);
		o->append(b);
	}
	if (t.reported == 0 && t.reported_paired == 0) o->append("No alignments\n");
	else if (t.reported_paired > 0 && t.reported == 0) { snprintf(b, sizeof(b), "Reported %llu paired-end alignments\n", (unsigned long long)(t.reported_paired >> 1)); o->append(b); }
	else if (t.reported_paired == 0) { snprintf(b, sizeof(b), "Reported %llu alignments\n", (unsigned long long)t.reported); o->append(b); }
	else { snprintf(b, sizeof(b), "Reported %llu paired-end alignments and %llu singleton alignments\n", (unsigned long long)(t.reported_paired >> 1), (unsigned long long)t.reported); o->append(b); }
}

/* ---- C entry points (include/bowtie_amd.h) ---------------------------------------------------- */
struct bt_reads {
	BtReadStream* s = nullptr;
	BtHostBatch batch;
	std::string err;
};

extern "C" int bt_reads_open(const char* spec, const bt_read_opts* opts, bt_reads** out)
{
	if (!spec || !opts || !out) return BT_ERR_ARG;
	if (opts->format < BT_FMT_FASTQ || opts->format > BT_FMT_TABBED || opts->trim5 < 0 || opts->trim3 < 0) return BT_ERR_ARG;
	/* the reference's tabbed reader has no usable integer-quality branch (pat.cpp:1079-1092 mangles the values after
	 * a space); interleaving is a FASTQ notion, one mate per stream */
	if (opts->format == BT_FMT_TABBED && (opts->qual_enc == BT_QUAL_INT || opts->qual_enc == BT_QUAL_INT_SOLEXA)) return BT_ERR_ARG;
	if ((opts->flags & BT_READ_INTERLEAVED) && (opts->format != BT_FMT_FASTQ || !(opts->flags & (BT_READ_MATE1 | BT_READ_MATE2)))) return BT_ERR_ARG;
	if (opts->format == BT_FMT_FASTA_CONT && (opts->cont_len < 1 || opts->cont_len >= 1024 || opts->cont_freq < 1)) return BT_ERR_ARG;
	bt_reads* r = new bt_reads();
	r->s = bt_io_open(spec, *opts, &r->err);
	*out = r;
	return BT_OK;
}
extern "C" int bt_reads_next(bt_reads* r, uint32_t max_reads, int threads, bt_read_batch* batch,
                             const char** names, const uint64_t** name_off)
{
	if (!r || !batch || max_reads == 0) return BT_ERR_ARG;
	r->err.clear();
	const int rc = bt_io_next(r->s, max_reads, threads, &r->batch, &r->err);
	if (rc != BT_OK) return rc;
	*batch = r->batch.view();
	if (names) *names = r->batch.names.data();
	if (name_off) *name_off = r->batch.name_off.data();
	return BT_OK;
}
extern "C" int bt_reads_raw(const bt_reads* r, const char** raw, const uint64_t** raw_off)
{
	if (!r || !raw || !raw_off || r->batch.raw_off.empty()) return BT_ERR_ARG;
	*raw = r->batch.raw.data(); *raw_off = r->batch.raw_off.data();
	return BT_OK;
}
extern "C" uint32_t bt_reads_paired_count(const bt_reads* r) { return r ? r->batch.n_paired : 0u; }
extern "C" const char* bt_reads_error(const bt_reads* r) { return r ? r->err.c_str() : ""; }
extern "C" void bt_reads_close(bt_reads* r)
{
	if (!r) return;
	bt_io_close(r->s);
	delete r;
}

static char* text_out(const std::string& s, size_t* len)
{
	char* t = (char*)malloc(s.size() + 1);
	if (!t) return nullptr;
	memcpy(t, s.data(), s.size()); t[s.size()] = 0;
	if (len) *len = s.size();
	return t;
}
static void refs_from(const char* const* refnames, const uint32_t* reflens, uint32_t n, BtRefNames* r)
{
	for (uint32_t i = 0; i < n; i++) { r->names.emplace_back(refnames && refnames[i] ? refnames[i] : ""); r->lens.push_back(reflens ? reflens[i] : 0); }
}
extern "C" int bt_format_hits(const bt_read_batch* reads, const char* names, const uint64_t* name_off,
                              const bt_hit_batch* hits, const char* const* refnames, const uint32_t* reflens,
                              uint32_t n_refs, const bt_out_opts* o, char** text, size_t* text_len, bt_out_tally* tally)
{
	if (!reads || !names || !name_off || !hits || !o || !text) return BT_ERR_ARG;
	BtRefNames refs; refs_from(refnames, reflens, n_refs, &refs);
	std::string s;
	bt_io_format(*reads, names, name_off, *hits, refs, *o, 0, reads->n_reads, &s, tally);
	*text = text_out(s, text_len);
	return *text ? BT_OK : BT_ERR_IO;
}
extern "C" int bt_format_sam_header(const char* const* refnames, const uint32_t* reflens, uint32_t n_refs,
                                    const bt_out_opts* o, const char* cmdline, const char* rgline,
                                    char** text, size_t* text_len)
{
	if (!o || !text) return BT_ERR_ARG;
	BtRefNames refs; refs_from(refnames, reflens, n_refs, &refs);
	std::string s;
	bt_io_sam_header(refs, *o, cmdline, rgline, &s);
	*text = text_out(s, text_len);
	return *text ? BT_OK : BT_ERR_IO;
}
extern "C" int bt_format_pairs(const bt_read_batch* r1, const char* names1, const uint64_t* name_off1,
                               const bt_read_batch* r2, const char* names2, const uint64_t* name_off2,
                               const bt_hit_batch* hits, const char* const* refnames, const uint32_t* reflens,
                               uint32_t n_refs, const bt_out_opts* o, char** text, size_t* text_len, bt_out_tally* tally)
{
	if (!r1 || !r2 || !names1 || !names2 || !name_off1 || !name_off2 || !hits || !o || !text || r1->n_reads != r2->n_reads) return BT_ERR_ARG;
	BtRefNames refs;
	for (uint32_t i = 0; i < n_refs; i++) { refs.names.emplace_back(refnames && refnames[i] ? refnames[i] : ""); refs.lens.push_back(reflens ? reflens[i] : 0); }
	std::string s;
	bt_io_format_pairs(*r1, names1, name_off1, *r2, names2, name_off2, *hits, refs, *o, 0, r1->n_reads, &s, tally);
	char* p = (char*)malloc(s.size() + 1);
	if (!p) return BT_ERR_ARG;
	memcpy(p, s.data(), s.size()); p[s.size()] = 0;
	*text = p;
	if (text_len) *text_len = s.size();
	return BT_OK;
}

extern "C" int bt_format_summary(const bt_out_tally* tally, char** text, size_t* text_len)
{
	if (!tally || !text) return BT_ERR_ARG;
	std::string s;
	bt_io_summary(*tally, &s);
	*text = text_out(s, text_len);
	return *text ? BT_OK : BT_ERR_IO;
}
extern "C" void bt_text_free(char* t) { free(t); }
