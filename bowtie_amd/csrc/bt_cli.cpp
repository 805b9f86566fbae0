/*
 * bowtie-amd -- command-line front end over the C ABI (include/bowtie_amd.h): the reference's
 * `bowtie [options]* -x <ebwt> <reads> [<hits>]` surface for unpaired, non---best alignment.
 *
 *   option table, defaults, validation   ebwt_search.cpp:144-254, 256-430, 590-925
 *   driver (index load, search, finish)  ebwt_search.cpp:2886-3290
 *   usage text                           ebwt_search.cpp:433-540
 *
 * Three host stages run concurrently over batches of reads: parse (bt_reads_next) -> search on the
 * GPU (bt_align_batch) -> format + write (bt_io_format), each batch in read order, so the output is
 * that of the reference run with -p 1.  Everything the search does happens behind the C ABI; this
 * file holds no alignment logic and has no CPU search path.
 */
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <unistd.h>
#include <vector>

#include <errno.h>
#include <limits.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>

#include <new>
#include <unordered_set>
#include <sys/mman.h>

#include "bt_io.h"

/* ---- where the large blocks come from ---------------------------------------------------------------------------------
 * A run over 192 M reads has thirteen batches of 12 M reads in flight: 65 GB of parsed reads, names, result arrays and SAM
 * text, nearly all of it touched once.  Out of the C library's allocator that is 16 M page faults on the parser's and the
 * formatter's threads, a third of a second of munmap() whenever a batch is let go, and seconds of the same inside exit().
 * Blocks of 4 MB and more are therefore mapped here, aligned to 2 MB and marked for transparent huge pages (a fault and an
 * unmap per 2 MB instead of per 4 KB; where the host has them switched off the mapping is an ordinary one).
 * BT_CLI_HUGEPAGES=0: everything from malloc() as before.  These replace the global operator new / delete of the program
 * (the library's C++ allocations included; its malloc()s are not touched). */
namespace bigmem {
constexpr size_t kHuge = (size_t)2u << 20, kMin = (size_t)4u << 20, kHdr = 64;
constexpr uint64_t kMagic0 = 0x62742d616d642d62ull, kMagic1 = 0x69672d626c6f636bull;
struct Hdr { uint64_t m0, base, len, m1; };
static int g_on = -1;
static size_t g_min = kMin;
/* (BT_CLI_BIG_MIN, bytes: the tests' way of putting small runs through these blocks) */
static bool on()
{
	if (g_on < 0) {
		const char* m = getenv("BT_CLI_BIG_MIN");
		if (m && atol(m) >= 4096) g_min = (size_t)atol(m);
		const char* e = getenv("BT_CLI_HUGEPAGES");
		g_on = e ? (atoi(e) != 0 ? 1 : 0) : 1;
	}
	return g_on == 1;
}
static void* map_block(size_t n)
{
	const size_t use = (n + kHdr + kHuge - 1) & ~(kHuge - 1), len = use + kHuge;
	if (use < n) return nullptr;
	void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (m == MAP_FAILED) return nullptr;
	const uintptr_t lo = (uintptr_t)m, a = (lo + kHuge - 1) & ~(uintptr_t)(kHuge - 1), end = lo + len;
	if (a > lo) munmap(m, a - lo);
	if (end > a + use) munmap((void*)(a + use), end - (a + use));
#ifdef MADV_HUGEPAGE
	(void)madvise((void*)a, use, MADV_HUGEPAGE);
#endif
	Hdr* h = (Hdr*)a;
	h->m0 = kMagic0; h->base = (uint64_t)a; h->len = (uint64_t)use; h->m1 = kMagic1;
	return (void*)(a + kHdr);
}
/* true: p was one of ours and is gone.  A pointer that sits 64 bytes past a 2 MB boundary and is not ours came from
 * malloc(): the 64 bytes before it are heap, readable, and do not hold both magic words and their own address. */
static bool unmap_block(void* p)
{
	if (((uintptr_t)p & (kHuge - 1)) != kHdr) return false;
	Hdr* h = (Hdr*)((uintptr_t)p - kHdr);
	if (h->m0 != kMagic0 || h->m1 != kMagic1 || h->base != (uint64_t)(uintptr_t)h) return false;
	munmap((void*)h, (size_t)h->len);
	return true;
}
}  /* namespace bigmem */

void* operator new(size_t n)
{
	if (n >= 4096 && bigmem::on() && n >= bigmem::g_min) { void* p = bigmem::map_block(n); if (p) return p; }
	for (;;) {
		void* p = malloc(n ? n : 1);
		if (p) return p;
		std::new_handler h = std::get_new_handler();
		if (!h) throw std::bad_alloc();
		h();
	}
}
void operator delete(void* p) noexcept { if (p && !bigmem::unmap_block(p)) free(p); }
void operator delete(void* p, size_t) noexcept { if (p && !bigmem::unmap_block(p)) free(p); }

namespace {

/* read batches (bt_io_set_allocator): the same blocks */
void* big_alloc(size_t bytes) { try { return ::operator new(bytes); } catch (...) { return nullptr; } }
void big_free(void* p) { ::operator delete(p); }

/* BT_CLI_PINNED_RESULTS=1: a batch's result arrays live in page-locked memory from the library (bt_host_alloc), so that the
 * copies back from the device are DMAs, not the runtime's staged copies into pageable memory.  Measured (GPU calls 9-12) it
 * changes nothing that can be seen -- the batches at the end of a run come back 0.2 s apart because that is how they complete,
 * not because of the copies -- while page-locking 0.5 GB per batch costs the reader's thread its time and 8 GB of such memory
 * is half a second of the process's exit: off by default.  Small arrays are never worth a page-locking call. */
int g_pin_results = -1;
std::mutex g_pin_m;
std::unordered_set<void*> g_pinned;
/* Once the input is exhausted the arrays of the batches still in flight are not handed back any more: un-locking them takes the
 * runtime's lock a dozen times per batch while the last batches are being written and the contexts let go, and the process is
 * about to leave (GPU call 11: 1.4 s of a 19.9 s run).  BT_CLI_TEARDOWN=1 frees everything by hand, as a leak check wants it. */
std::atomic<bool> g_leave_pinned(false);
template <typename T> struct PinAlloc {
	using value_type = T;
	PinAlloc() = default;
	template <typename U> PinAlloc(const PinAlloc<U>&) {}
	T* allocate(size_t n)
	{
		const size_t bytes = n * sizeof(T);
		if (g_pin_results < 0) { const char* e = getenv("BT_CLI_PINNED_RESULTS"); g_pin_results = e && atoi(e) != 0 ? 1 : 0; }
		if (g_pin_results == 1 && bytes >= ((size_t)1u << 20)) {
			void* p = bt_host_alloc(bytes);
			if (p) { std::lock_guard<std::mutex> l(g_pin_m); g_pinned.insert(p); return (T*)p; }
		}
		return (T*)::operator new(bytes);
	}
	void deallocate(T* p, size_t)
	{
		{
			std::unique_lock<std::mutex> l(g_pin_m);
			auto it = g_pinned.find((void*)p);
			if (it != g_pinned.end()) { g_pinned.erase(it); l.unlock(); if (!g_leave_pinned.load()) bt_host_free((void*)p); return; }
		}
		::operator delete((void*)p);
	}
	template <typename U> bool operator==(const PinAlloc<U>&) const { return true; }
	template <typename U> bool operator!=(const PinAlloc<U>&) const { return false; }
};
template <typename T> using PinVec = std::vector<T, PinAlloc<T>>;

struct Options {
	bt_policy pol;
	bt_read_opts rd;
	bt_out_opts out;
	std::string index, reads, hits_file, rg_id;
	std::string dump_al, dump_un, dump_max;   /* --al / --un / --max */
	std::vector<std::string> rg_fields;
	int threads = 0, offrate = -1, inflight = 2;      /* threads 0 = pick from the host */
	bool no_stream = false, stream = false;
	bool large_index = false;                /* --large-index: the reference's wrapper then runs bowtie-align-l, which reads only the .ebwtl files (bowtie:64-65) */
	std::vector<int> devices;                /* GPUs the batches are dealt to (default: 0) */
	bool quiet = false, timing = false, sam_nohead = false, tryhard = false, maxbts_set = false, paired = false;
	bool best_given = false;     /* --best itself: what clears the reference's useV1 (ebwt_search.cpp:776); -v 3 / -M only set `stateful` */
	std::string mates1, mates2;
	std::string tab12, ileaved;              /* --12 / --interleaved file lists */
	std::string quals, quals1, quals2;       /* -Q / --Q1 / --Q2 */
	bool interleaved = false;
	bool suppress_set = false, int_quals = false;
	/* reads per batch = per launch.  With carry-over a read may ride along for twelve launches; the heaviest reads need
	 * ~5 s of them, so a launch must last long enough: at 4 M reads (0.34 s) every launch ended up waiting for the stragglers
	 * of the batch twelve launches back -- 3.8 M reads/s GPU-side on 192 M reads, 5.3 M file to file at 8 M, 5.6 M at 16 M
	 * (where parsing bounds it: profiles/r5/call12_cli_batch_SUMMARY.txt).  8 M keeps a batch at ~2.4 GB of host memory.
	 * Round 6 (the reader no longer bounds anything): 192 M reads in 28.8-30.0 s at 8 M, 25.5 s at 12 M, 27.1 s at 16 M per batch
	 * (profiles/r6/call6_*) -- a streamed run on a host with the memory for it (thirteen batches in flight: 45 GB) takes 12 M
	 * unless --batch says otherwise */
	uint32_t batch_reads = 8u << 20;
	bool batch_set = false;
	std::string cmdline;
};

[[noreturn]] void die(const char* fmt, ...)
{
	va_list ap; va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fputc('\n', stderr);
	exit(1);
}

void usage(FILE* o)
{
	fputs(
	    "Usage: \n"
	    "bowtie-amd [options]* -x <ebwt> <s> [<hit>]\n\n"
	    "  <ebwt>  Index filename prefix (minus trailing .X.ebwt).\n"
	    "  <s>     Comma-separated list of files containing unpaired reads, or the\n"
	    "          sequences themselves, if -c is set.  Specify \"-\" for stdin.\n"
	    "  <hit>   File to write hits to (default: stdout)\n"
	    "Input:\n"
	    "  -q                 query input files are FASTQ .fq/.fastq (default)\n"
	    "  -f                 query input files are (multi-)FASTA .fa/.mfa\n"
	    "  -F <len>,<int>     query input files are FASTA whose reads are the <len>-mers at every\n"
	    "                     <int>-th position of each record\n"
	    "  -r                 query input files are raw one-sequence-per-line\n"
	    "  --12 <files>       tab-delimited reads, one per line: name, seq, quals -- or, for a pair, name,\n"
	    "                     seq1, quals1, seq2, quals2 (a file holds one kind); needs --best\n"
	    "  --interleaved <files>  FASTQ in which first and second mates alternate; needs --best\n"
	    "  -c                 query sequences given on cmd line (as <s>)\n"
	    "  -s/--skip <int>    skip the first <int> reads in the input\n"
	    "  -u/--qupto <int>   stop after first <int> reads (excl. skipped reads)\n"
	    "  -5/--trim5 <int>   trim <int> bases from 5' (left) end of reads\n"
	    "  -3/--trim3 <int>   trim <int> bases from 3' (right) end of reads\n"
	    "  --phred33-quals    input quals are Phred+33 (default)\n"
	    "  --phred64-quals    input quals are Phred+64 (same as --solexa1.3-quals)\n"
	    "  --solexa-quals     input quals are from GA Pipeline ver. < 1.3\n"
	    "  --solexa1.3-quals  input quals are from GA Pipeline ver. >= 1.3\n"
	    "  --integer-quals    qualities are given as space-separated integers (not ASCII)\n"
	    "  --large-index      force usage of a 'large' index, even if a small one is present\n"
	    "Alignment:\n"
	    "  -v <int>           report end-to-end hits w/ <=v mismatches; ignore qualities (0-2)\n"
	    "    or\n"
	    "  -n/--seedmms <int> max mismatches in seed (can be 0-3, default: -n 2)\n"
	    "  -e/--maqerr <int>  max sum of mismatch quals across alignment for -n (def: 70)\n"
	    "  -l/--seedlen <int> seed length for -n (default: 28)\n"
	    "  --nomaqround       disable Maq-like quality rounding for -n (nearest 10 <= 30)\n"
	    "  --nofw/--norc      do not align to forward/reverse-complement reference strand\n"
	    "  --maxbts <int>     max # backtracks for -n 2/3 (default: 125)\n"
	    "  -y/--tryhard       try hard to find valid alignments, at the expense of speed\n"
	    "Reporting:\n"
	    "  -k <int>           report up to <int> good alignments per read (default: 1)\n"
	    "  -a/--all           report all alignments per read (much slower than low -k)\n"
	    "  -m <int>           suppress all alignments if > <int> exist (def: no limit)\n"
	    "Output:\n"
	    "  -t/--time          print wall-clock time taken by search phases\n"
	    "  -B/--offbase <int> leftmost ref offset = <int> in bowtie output (default: 0)\n"
	    "  --quiet            print nothing but the alignments\n"
	    "  --refidx           refer to ref. seqs by 0-based index rather than name\n"
	    "  --al <fname>       write aligned reads to file <fname>\n"
	    "  --un <fname>       write unaligned reads to file <fname>\n"
	    "  --max <fname>      write reads exceeding -m limit to file <fname>\n"
	    "  --fullref          write entire ref name (default: only up to 1st space)\n"
	    "  --suppress <cols>  suppresses given columns (comma-delim'ed) in default output\n"
	    "  --cost / --showseed  extra columns in default output\n"
	    "SAM:\n"
	    "  -S/--sam           write hits in SAM format\n"
	    "  --mapq <int>       default mapping quality (MAPQ) to print for SAM alignments\n"
	    "  --sam-nohead       supppress header lines (starting with @) for SAM output\n"
	    "  --sam-nosq         supppress @SQ header lines for SAM output\n"
	    "  --sam-RG <text>    add <text> (usually \"lab=value\") to @RG line of SAM header\n"
	    "  --sam-no-qname-trunc  do not cut read names at the first whitespace\n"
	    "  --no-unal          suppress SAM records for unaligned reads\n"
	    "Performance:\n"
	    "  -o/--offrate <int> override offrate of index; must be >= index's offrate\n"
	    "  -p/--threads <int> number of host threads for parsing and formatting (default: all, up to 32);\n"
	    "                     the output order is that of the input whatever the value\n"
	    "  --device <list>    GPU(s) to run on, e.g. 0,1,2,3: index replicated, batches dealt out (default: 0)\n"
	    "  --batch <int>      reads per GPU batch (default: 8388608)\n"
	    "  --inflight <int>   batches searched concurrently, each on its own stream (default: 2)\n"
	    "  --stream           the default for unpaired reads without --best: batches stream through one\n"
	    "                     context per GPU, reads still running when a batch ends are carried into the next\n"
	    "  --no-stream        search --inflight whole batches side by side instead\n"
	    "Other:\n"
	    "  --seed <int>       seed for random number generator\n"
	    "  --version          print version information and quit\n"
	    "  -h/--help          print this usage message\n"
	    "Paired-end (without --best: the reference's PairedBWAlignerV1, its default; with --best: PairedBWAlignerV2):\n"
	    "  -1 <m1> -2 <m2>    files with #1 and #2 mates (comma-separated lists; same format options as <s>)\n"
	    "  -I/--minins <int>  minimum insert size (default: 0)\n"
	    "  -X/--maxins <int>  maximum insert size (default: 250)\n"
	    "  --fr/--rf/--ff     -1, -2 mates align fw/rev, rev/fw, fw/fw (default: --fr)\n"
	    "  --pairtries <int>  max # anchors tried per pair (default: 100)\n"
	    "  --allow-contain    one mate alignment may contain the other\n"
	    "  --12 <f> / --interleaved <f>   pairs (--12: also unpaired reads, or both) from one tab-delimited / FASTQ file;\n"
	    "                     - = standard input (spooled to a temporary file)\n"
	    "Not in this build: -z, --mm / --shmem, indexes of 2^32-1 rows or more\n",
	    o);
}

long parse_int(const char* s, long lo, const char* msg)
{
	char* e = nullptr; errno = 0;
	const long v = strtol(s ? s : "", &e, 10);
	if (!s || e == s || *e || errno || v < lo || v > INT_MAX) { fprintf(stderr, "%s\n", msg); usage(stderr); exit(1); }
	return v;
}

struct LongOpt { const char* name; int has_arg; int id; };
enum {
	O_SOLEXA = 256, O_PHRED64, O_PHRED33, O_SEED, O_MAXBTS, O_QUIET, O_REFIDX, O_FULLREF, O_NOMAQ, O_NOFW, O_NORC,
	O_SAM_NOHEAD, O_SAM_NOSQ, O_SAM_RG, O_SAM_NOTRUNC, O_NO_UNAL, O_MAPQ, O_SUPPRESS, O_COST, O_SHOWSEED, O_VERSION,
	O_USAGE, O_BEST, O_STRATA, O_FF, O_FR, O_RF, O_PAIRTRIES, O_ALLOW_CONTAIN, O_DEVICE, O_BATCH, O_INFLIGHT, O_NOSTREAM, O_STREAM, O_WRAPPER, O_AL, O_UN, O_MAX, O_INTQUALS, O_TAB12, O_ILEAVED, O_QUALS1, O_QUALS2, O_LARGE_INDEX, O_IGNORED, O_IGNORED_ARG, O_UNSUPPORTED, O_UNSUPPORTED_ARG
};
const LongOpt LONGS[] = {
	{"all", 0, 'a'}, {"solexa-quals", 0, O_SOLEXA}, {"time", 0, 't'}, {"trim3", 1, '3'}, {"trim5", 1, '5'}, {"seed", 1, O_SEED},
	{"qupto", 1, 'u'}, {"offrate", 1, 'o'}, {"version", 0, O_VERSION}, {"maqerr", 1, 'e'}, {"seedlen", 1, 'l'}, {"seedmms", 1, 'n'},
	{"help", 0, 'h'}, {"threads", 1, 'p'}, {"khits", 1, 'k'}, {"mhits", 1, 'm'}, {"nomaqround", 0, O_NOMAQ}, {"refidx", 0, O_REFIDX},
	{"maxbts", 1, O_MAXBTS}, {"nofw", 0, O_NOFW}, {"norc", 0, O_NORC}, {"offbase", 1, 'B'}, {"tryhard", 0, 'y'}, {"skip", 1, 's'},
	{"phred33-quals", 0, O_PHRED33}, {"phred64-quals", 0, O_PHRED64}, {"solexa1.3-quals", 0, O_PHRED64}, {"fullref", 0, O_FULLREF},
	{"usage", 0, O_USAGE}, {"sam", 0, 'S'}, {"sam-no-qname-trunc", 0, O_SAM_NOTRUNC}, {"sam-nohead", 0, O_SAM_NOHEAD},
	{"sam-nosq", 0, O_SAM_NOSQ}, {"sam-noSQ", 0, O_SAM_NOSQ}, {"sam-RG", 1, O_SAM_RG}, {"suppress", 1, O_SUPPRESS}, {"mapq", 1, O_MAPQ},
	{"cost", 0, O_COST}, {"showseed", 0, O_SHOWSEED}, {"no-unal", 0, O_NO_UNAL}, {"quiet", 0, O_QUIET}, {"device", 1, O_DEVICE},
	{"batch", 1, O_BATCH}, {"inflight", 1, O_INFLIGHT}, {"no-stream", 0, O_NOSTREAM}, {"stream", 0, O_STREAM}, {"wrapper", 1, O_WRAPPER},
	/* accepted and without effect here (host-memory / CPU-threading knobs of the reference) */
	{"reads-per-batch", 1, O_IGNORED_ARG}, {"chunkmbs", 1, O_IGNORED_ARG}, {"chunksz", 1, O_IGNORED_ARG}, {"chunkverbose", 0, O_IGNORED},
	{"verbose", 0, O_IGNORED}, {"startverbose", 0, O_IGNORED}, {"sanity", 0, O_IGNORED}, {"reorder", 0, O_IGNORED},
	{"thread-ceiling", 1, O_IGNORED_ARG}, {"thread-piddir", 1, O_IGNORED_ARG}, {"mm", 0, O_IGNORED}, {"shmem", 0, O_IGNORED},
	{"mmsweep", 0, O_IGNORED}, {"prewidth", 1, O_IGNORED_ARG}, {"cachelim", 1, O_IGNORED_ARG}, {"cachesz", 1, O_IGNORED_ARG},
	{"pause", 0, O_IGNORED}, {"stats", 0, O_IGNORED}, {"reportopps", 0, O_UNSUPPORTED}, {"mixthresh", 1, O_UNSUPPORTED_ARG}, {"stateful", 0, O_UNSUPPORTED}, {"large-index", 0, O_LARGE_INDEX},
	/* the best-first engine and everything that needs it */
	{"best", 0, O_BEST}, {"better", 0, O_UNSUPPORTED}, {"oldbest", 0, O_UNSUPPORTED}, {"strata", 0, O_STRATA},
	{"minins", 1, 'I'}, {"maxins", 1, 'X'}, {"ff", 0, O_FF}, {"fr", 0, O_FR},
	{"rf", 0, O_RF}, {"12", 1, O_TAB12}, {"interleaved", 1, O_ILEAVED}, {"pairtries", 1, O_PAIRTRIES},
	{"integer-quals", 0, O_INTQUALS}, {"quals", 1, 'Q'}, {"Q1", 1, O_QUALS1}, {"Q2", 1, O_QUALS2},
	{"al", 1, O_AL}, {"un", 1, O_UN}, {"max", 1, O_MAX}, {"phased", 0, O_UNSUPPORTED},
	{"strandfix", 0, O_IGNORED}, {"pev2", 0, O_UNSUPPORTED}, {"reportse", 0, O_UNSUPPORTED}, {"hadoopout", 0, O_UNSUPPORTED},
	{"partition", 1, O_UNSUPPORTED_ARG}, {"range", 0, O_UNSUPPORTED}, {"isarate", 1, O_UNSUPPORTED_ARG}, {"allow-contain", 0, O_ALLOW_CONTAIN},
	{"orig", 1, O_UNSUPPORTED_ARG}, {"filepar", 0, O_UNSUPPORTED}, {"noreconcile", 0, O_UNSUPPORTED},
	{nullptr, 0, 0}
};
/* short options taking an argument */
const char* SHORT_ARG = "us35oenlpkmMBxvF12IXQ";
const char* SHORT_UNSUPPORTED_ARG = "w";
const char* SHORT_UNSUPPORTED = "bz";

/* --12: does the file's first record carry a second end?  (The reference decides per record and aligns pairs and
 * single reads side by side, pat.h:990-995; here a file is one or the other, checked batch by batch.) */
std::vector<std::string> split_commas(const std::string& s)
{
	std::vector<std::string> v;
	size_t b = 0;
	while (b <= s.size()) {
		const size_t e = s.find(',', b);
		const std::string t = s.substr(b, e == std::string::npos ? std::string::npos : e - b);
		if (!t.empty()) v.push_back(t);
		if (e == std::string::npos) break;
		b = e + 1;
	}
	return v;
}

/* the i-th quality file goes with the i-th read file (CFilePatternSource::open, pat.cpp:333-347) */
void drop_reads_without_quals(std::string* reads, const std::string& quals)
{
	if (quals.empty() || reads->empty()) return;
	const std::vector<std::string> r = split_commas(*reads), q = split_commas(quals);
	std::string kept;
	for (size_t i = 0; i < r.size(); i++) {
		bool ok = true;
		if (i < q.size() && q[i] != "-") {
			FILE* f = fopen(q[i].c_str(), "rb");
			if (f) fclose(f);
			else { fprintf(stderr, "Warning: Could not open quality file \"%s\" for reading; skipping...\n", q[i].c_str()); ok = false; }
		}
		/* a read file that is left out keeps its place in the list, marked (bt_io.cpp: st_open_next): the run then ends
		 * the way the reference's does when the files left cannot be read */
		if (!kept.empty()) kept.push_back(',');
		kept.append(ok ? r[i] : std::string("\x01") + r[i]);
	}
	*reads = kept;
}

/* --12 - / --interleaved -: the two mate streams each read the input from the start, which a pipe cannot give them;
 * standard input goes to a temporary file (removed at exit) and its name stands in */
std::string g_spooled;
void remove_spooled() { if (!g_spooled.empty()) unlink(g_spooled.c_str()); }
std::string spool_stdin()
{
	if (!g_spooled.empty()) return g_spooled;                  /* "-" twice: one stream of bytes, one file */
	const char* td = getenv("TMPDIR");
	std::string path = std::string(td && *td ? td : "/tmp") + "/bowtie_amd_stdin_XXXXXX";
	std::vector<char> buf(path.begin(), path.end()); buf.push_back(0);
	const int fd = mkstemp(buf.data());
	if (fd < 0) die("Error: could not create a temporary file for standard input");
	g_spooled = buf.data();
	atexit(remove_spooled);
	std::vector<char> blk(1 << 20);
	for (;;) {
		const ssize_t n = read(0, blk.data(), blk.size());
		if (n < 0) die("Error: reading standard input failed");
		if (n == 0) break;
		for (ssize_t off = 0; off < n;) { const ssize_t w = write(fd, blk.data() + off, (size_t)(n - off)); if (w <= 0) die("Error: writing %s failed", g_spooled.c_str()); off += w; }
	}
	close(fd);
	return g_spooled;
}

void parse_args(int argc, char** argv, Options* O)
{
	bt_policy_default(&O->pol);
	memset(&O->rd, 0, sizeof(O->rd));
	memset(&O->out, 0, sizeof(O->out));
	O->rd.format = BT_FMT_FASTQ;
	O->out.mapq = 255;
	std::vector<std::string> pos;
	for (int i = 0; i < argc; i++) { if (i) O->cmdline.push_back(' '); O->cmdline.append(argv[i]); }
	for (int i = 1; i < argc; i++) {
		const char* a = argv[i];
		if (a[0] != '-' || a[1] == 0) { pos.emplace_back(a); continue; }
		int id = 0; const char* val = nullptr;
		if (a[1] == '-') {
			if (a[2] == 0) { for (i++; i < argc; i++) pos.emplace_back(argv[i]); break; }
			std::string name(a + 2); std::string inl; bool has_inl = false;
			const size_t eq = name.find('=');
			if (eq != std::string::npos) { inl = name.substr(eq + 1); name.resize(eq); has_inl = true; }
			const LongOpt* lo = nullptr;
			for (const LongOpt* p = LONGS; p->name; p++) if (name == p->name) { lo = p; break; }
			if (!lo) { fprintf(stderr, "bowtie-amd: unrecognized option '--%s'\n", name.c_str()); usage(stderr); exit(1); }
			id = lo->id;
			if (lo->has_arg) {
				static std::string keep;
				if (has_inl) { keep = inl; val = keep.c_str(); }
				else { if (i + 1 >= argc) die("bowtie-amd: option '--%s' requires an argument", name.c_str()); val = argv[++i]; }
			}
			if (id == O_UNSUPPORTED || id == O_UNSUPPORTED_ARG)
				die("Error: --%s selects a part of the reference that this build does not have (DESIGN.md 1: what is not built)", name.c_str());
			if (id == O_IGNORED || id == O_IGNORED_ARG) continue;
		} else {
			/* a short option with its value (attached or next argument), or a bundle of flags */
			const char c = a[1];
			if (strchr(SHORT_UNSUPPORTED_ARG, c) || strchr(SHORT_UNSUPPORTED, c))
				die("Error: -%c selects a part of the reference that this build does not have (DESIGN.md 1: what is not built)", c);
			if (strchr(SHORT_ARG, c)) {
				if (a[2]) val = a + 2;
				else { if (i + 1 >= argc) die("bowtie-amd: option requires an argument -- '%c'", c); val = argv[++i]; }
				id = c;
			} else {
				for (const char* p = a + 1; *p; p++) {
					switch (*p) {
					case 'f': O->rd.format = BT_FMT_FASTA; break;
					case 'q': O->rd.format = BT_FMT_FASTQ; break;
					case 'r': O->rd.format = BT_FMT_RAW; break;
					case 'c': O->rd.format = BT_FMT_CMDLINE; break;
					case 'a': O->pol.all_hits = 1; break;
					case 't': O->timing = true; break;
					case 'y': O->tryhard = true; break;
					case 'S': O->out.sam = 1; break;
					case 'h': usage(stdout); exit(0);
					case 'C': die("Error: -C specified but Bowtie no longer supports colorspace.");
					default: fprintf(stderr, "bowtie-amd: invalid option -- '%c'\n", *p); usage(stderr); exit(1);
					}
				}
				continue;
			}
		}
		switch (id) {
		case 'a': O->pol.all_hits = 1; break;
		case 't': O->timing = true; break;
		case 'y': O->tryhard = true; break;
		case 'S': O->out.sam = 1; break;
		case 'h': case O_USAGE: usage(stdout); exit(0);
		case 'u': O->rd.upto = (uint64_t)parse_int(val, 1, "-u/--qupto arg must be at least 1"); break;
		case 's': O->rd.skip = (uint64_t)parse_int(val, 0, "-s arg must be positive"); break;
		case '3': O->rd.trim3 = (int32_t)parse_int(val, 0, "-3/--trim3 arg must be at least 0"); break;
		case '5': O->rd.trim5 = (int32_t)parse_int(val, 0, "-5/--trim5 arg must be at least 0"); break;
		case 'o': O->offrate = (int)parse_int(val, 1, "-o/--offrate arg must be at least 1"); break;
		case 'e': O->pol.qual_thresh = (int32_t)parse_int(val, 1, "-e/--err arg must be at least 1"); break;
		case 'n': O->pol.mode = BT_MODE_N; O->pol.mms = (int32_t)parse_int(val, 0, "-n/--seedmms arg must be at least 0"); break;
		case 'l': O->pol.seed_len = (int32_t)parse_int(val, 5, "-l/--seedlen arg must be at least 5"); break;
		case 'v': break;
		case 'p': O->threads = (int)parse_int(val, 1, "-p/--threads arg must be at least 1"); break;
		case 'k': O->pol.khits = (uint32_t)parse_int(val, 1, "-k arg must be at least 1"); break;
		case 'M': O->pol.sample_max = 1;    /* falls through: -M <n> is -m <n> plus sampling (ebwt_search.cpp:721-725) */
		case 'm': O->pol.mhits = (uint32_t)parse_int(val, 1, "-m arg must be at least 1"); break;
		case '1': if (!O->mates1.empty()) O->mates1.push_back(','); O->mates1.append(val); break;
		case '2': if (!O->mates2.empty()) O->mates2.push_back(','); O->mates2.append(val); break;
		case 'Q': if (!O->quals.empty()) O->quals.push_back(','); O->quals.append(val); break;
		case O_QUALS1: if (!O->quals1.empty()) O->quals1.push_back(','); O->quals1.append(val); break;
		case O_QUALS2: if (!O->quals2.empty()) O->quals2.push_back(','); O->quals2.append(val); break;
		case O_TAB12: if (!O->tab12.empty()) O->tab12.push_back(','); O->tab12.append(val); O->rd.format = BT_FMT_TABBED; break;
		case O_ILEAVED: if (!O->ileaved.empty()) O->ileaved.push_back(','); O->ileaved.append(val); O->rd.format = BT_FMT_FASTQ; break;
		case 'I': O->pol.min_ins = (int32_t)parse_int(val, 0, "-I arg must be positive"); break;
		case 'X': O->pol.max_ins = (int32_t)parse_int(val, 1, "-X arg must be at least 1"); break;
		case O_FF: O->pol.mate1_fw = 1; O->pol.mate2_fw = 1; break;
		case O_RF: O->pol.mate1_fw = 0; O->pol.mate2_fw = 1; break;
		case O_FR: O->pol.mate1_fw = 1; O->pol.mate2_fw = 0; break;
		case O_PAIRTRIES: O->pol.pair_tries = (int32_t)parse_int(val, 1, "--pairtries arg must be at least 1"); break;
		case O_ALLOW_CONTAIN: O->pol.allow_contain = 1; break;
		case O_BEST: O->pol.best = 1; O->best_given = true; break;
		case O_STRATA: O->pol.strata = 1; break;
		case 'B': O->out.off_base = (int32_t)parse_int(val, -999999, "-B/--offbase arg must be at least -999999"); break;
		case 'x': O->index = val; break;
		case 'F': {
			/* parsePair<size_t>(optarg, ','): two plain numbers, read length and interval */
			char* e = nullptr;
			const long k = strtol(val, &e, 10);
			if (e == val || *e != ',' || k < 1 || k >= 1024) die("Error: -F takes <length>,<interval> (two numbers, length < 1024): %s", val);
			const char* q = e + 1;
			const long iv = strtol(q, &e, 10);
			if (e == q || *e || iv < 1) die("Error: -F takes <length>,<interval> (two numbers, length < 1024): %s", val);
			O->rd.format = BT_FMT_FASTA_CONT; O->rd.cont_len = (uint32_t)k; O->rd.cont_freq = (uint32_t)iv;
			break;
		}
		case O_SOLEXA: O->rd.qual_enc = BT_QUAL_SOLEXA64; break;
		case O_PHRED64: O->rd.qual_enc = BT_QUAL_PHRED64; break;
		case O_PHRED33: O->rd.qual_enc = BT_QUAL_PHRED33; break;
		case O_SEED: O->rd.seed = (uint32_t)parse_int(val, 0, "--seed arg must be at least 0"); break;
		case O_MAXBTS: O->pol.max_bts = (int32_t)parse_int(val, 0, "--maxbts must be positive"); O->maxbts_set = true; break;
		case O_QUIET: O->quiet = true; break;
		case O_REFIDX: O->out.ref_idx = 1; break;
		case O_FULLREF: O->out.full_ref = 1; break;
		case O_NOMAQ: O->pol.maq_round = 0; break;
		case O_NOFW: O->pol.nofw = 1; break;
		case O_NORC: O->pol.norc = 1; break;
		case O_SAM_NOHEAD: O->sam_nohead = true; break;
		case O_SAM_NOSQ: O->out.sam_nosq = 1; break;
		case O_SAM_NOTRUNC: O->out.no_qname_trunc = 1; break;
		case O_NO_UNAL: O->out.no_unal = 1; break;
		case O_MAPQ: O->out.mapq = (int32_t)parse_int(val, 0, "--mapq must be positive"); break;
		case O_COST: O->out.print_cost = 1; break;
		case O_SHOWSEED: O->out.show_seed = 1; break;
		case O_DEVICE: {
			/* "--device 2" or "--device 0,1,2,3": the index is replicated on every listed GPU and the
			 * batches are dealt to them; the output order does not depend on the list */
			O->devices.clear();
			const char* p = val;
			while (*p) {
				char* e = nullptr;
				const long d = strtol(p, &e, 10);
				if (e == p || d < 0 || d > 1023 || (*e && *e != ',')) die("Error: bad --device list: %s", val);
				O->devices.push_back((int)d);
				p = (*e == ',') ? e + 1 : e;
			}
			if (O->devices.empty()) die("Error: bad --device list: %s", val);
			break;
		}
		case O_BATCH: O->batch_reads = (uint32_t)parse_int(val, 1, "--batch arg must be at least 1"); O->batch_set = true; break;
		case O_NOSTREAM: O->no_stream = true; break;
		case O_STREAM: O->stream = true; break;
		case O_INFLIGHT: O->inflight = (int)parse_int(val, 1, "--inflight arg must be at least 1"); if (O->inflight > 4) O->inflight = 4; break;
		case O_WRAPPER: break;
		case O_LARGE_INDEX: O->large_index = true; break;
		case O_INTQUALS: O->int_quals = true; break;
		case O_AL: O->dump_al = val; break;
		case O_UN: O->dump_un = val; break;
		case O_MAX: O->dump_max = val; break;
		case O_VERSION: printf("bowtie-amd (%s), output-compatible with bowtie-align-s version 1.3.1\n", bt_version()); exit(0);
		case O_SAM_RG: {
			/* "ID:x" opens the @RG line, later fields are tab-appended (ebwt_search.cpp ARG_SAM_RG) */
			O->rg_fields.emplace_back(val);
			break;
		}
		case O_SUPPRESS: {
			const char* p = val;
			while (*p) {
				char* e = nullptr;
				const long f = strtol(p, &e, 10);
				if (e == p || f < 1 || f > 64) die("Error: bad --suppress field: %s", val);
				O->out.suppress |= 1ull << (f - 1);
				O->suppress_set = true;
				p = (*e == ',') ? e + 1 : e;
				if (*e && *e != ',') die("Error: bad --suppress field: %s", val);
			}
			break;
		}
		default: break;
		}
		if (id == 'v') {
			O->pol.mode = BT_MODE_V; O->pol.mms = (int32_t)parse_int(val, 0, "-v arg must be at least 0");
			if (O->pol.mms > 3) die("-v arg must be at most 3");
		}
	}
	/* what sends the reference to its stateful best-first workers (ebwt_search.cpp:851-853, 877-887) */
	if (O->pol.mode == BT_MODE_V && O->pol.mms == 3) O->pol.best = 1;
	if (!O->pol.best && O->pol.sample_max) {
		if (!O->quiet) fprintf(stderr, "Warning: -M was specified w/o --best; automatically enabling --best\n");
		O->pol.best = 1;
	}
	if (O->pol.strata && !O->pol.best) die("--strata must be combined with --best");
	if (O->pol.strata && !O->pol.all_hits && O->pol.khits == 1 && O->pol.mhits == 0xffffffffu)
		die("--strata has no effect unless combined with -m, -a, or -k N where N > 1");
	/* --maxbts: 125 for the phase programs, 800 for the best-first workers (ebwt_search.cpp:185-186) */
	if (O->pol.best && !O->maxbts_set) O->pol.max_bts = 800;
	/* the insert-size limits as resolveOutstandingInRef sees them (aligner.h:1921-1935): reduced by the
	 * trimming at the outer ends of the fragment (a mate shorter than its trims is skipped anyway) */
	{
		int mn = O->pol.min_ins, mx = O->pol.max_ins;
		const int o1 = O->pol.mate1_fw ? O->rd.trim5 : O->rd.trim3, o2 = O->pol.mate2_fw ? O->rd.trim3 : O->rd.trim5;
		mn = mn - o1 < 0 ? 0 : mn - o1; mx = mx - o1 < 0 ? 0 : mx - o1;
		mn = mn - o2 < 0 ? 0 : mn - o2; mx = mx - o2 < 0 ? 0 : mx - o2;
		O->pol.min_ins = mn; O->pol.max_ins = mx;
	}
	if (O->pol.mode == BT_MODE_N && O->pol.mms > 3) die("-n/--seedmms arg must be at most 3");
	/* positionals: [<ebwt>] <reads> [<hits>] (ebwt_search.cpp:2930-2975) */
	size_t pi = 0;
	if (O->index.empty()) {
		if (pi >= pos.size()) { fprintf(stderr, "No index, query, or output file specified!\n"); usage(stderr); exit(1); }
		O->index = pos[pi++];
	}
	const bool one_file = !O->tab12.empty() || !O->ileaved.empty();
	if (one_file) {
		/* --12 / --interleaved (ebwt_search.cpp:626-627): one file carries both mates.  Whether its records are pairs
		 * or not, the reference then runs its stateful aligners (:3001-3002) */
		if (!O->mates1.empty() || !O->mates2.empty() || (!O->tab12.empty() && !O->ileaved.empty()))
			die("Error: --12 / --interleaved cannot be combined with -1/-2 or with each other in this build");
		/* both mate streams open the file: standard input is spooled to a temporary file first */
		{
			std::string& spec = O->tab12.empty() ? O->ileaved : O->tab12;
			std::string out;
			for (const std::string& f : split_commas(spec)) {
				if (!out.empty()) out.push_back(',');
				out.append(f == "-" ? spool_stdin() : f);
			}
			spec = out;
		}
		if (!O->ileaved.empty()) { O->mates1 = O->mates2 = O->ileaved; O->interleaved = true; }
		else O->mates1 = O->mates2 = O->tab12;      /* a --12 file may hold pairs, unpaired reads or both (pat.cpp:977-1127) */
	}
	O->paired = !O->mates1.empty() || !O->mates2.empty();
	if (O->paired) {
		/* ebwt_search.cpp:855-860 */
		const size_t c1 = (size_t)std::count(O->mates1.begin(), O->mates1.end(), ',') + (O->mates1.empty() ? 0 : 1);
		const size_t c2 = (size_t)std::count(O->mates2.begin(), O->mates2.end(), ',') + (O->mates2.empty() ? 0 : 1);
		if (c1 != c2)
			die("Error: %zu mate files/sequences were specified with -1, but %zu\nmate files/sequences were specified with -2.  The same number of mate files/\nsequences must be specified with -1 and -2.", c1, c2);
		if (!O->best_given) {
			/* the reference's default paired-end aligner, PairedBWAlignerV1 (useV1, ebwt_search.cpp:232): only --best
			 * itself selects PairedBWAlignerV2 (:776) -- -v 3 and -M make the run stateful but leave useV1 alone */
			O->pol.pe_v1 = 1;
			if (!O->maxbts_set) O->pol.max_bts = 800;              /* every stateful aligner: ebwt_search.cpp:185-186, 2644, 2670 */
		}
	} else {
	if (pi >= pos.size()) { fprintf(stderr, "No query or output file specified!\n"); usage(stderr); exit(1); }
	O->reads = pos[pi++];
	}
	if (!O->quals.empty() || !O->quals1.empty() || !O->quals2.empty()) {
		/* -Q / --Q1 / --Q2: the reference hands the quality files to its FASTA reader only (ebwt_search.cpp:2925-2927),
		 * which opens each next to its read file and never reads from it (pat.cpp:333-347): a quality file that cannot
		 * be opened takes its read file out of the run, nothing else happens */
		if (O->rd.format != BT_FMT_FASTA) die("Error: -Q/--Q1/--Q2 go with -f; with other formats the reference reads qualities as integers, which this build does with --integer-quals");
		drop_reads_without_quals(&O->reads, O->quals);
		drop_reads_without_quals(&O->mates1, O->quals1);
		drop_reads_without_quals(&O->mates2, O->quals2);
	}
	if (pi < pos.size()) O->hits_file = pos[pi++];
	if (pi < pos.size()) { fprintf(stderr, "Extra parameter(s) specified: "); for (; pi < pos.size(); pi++) fprintf(stderr, "\"%s\"%s", pos[pi].c_str(), pi + 1 < pos.size() ? ", " : "\n"); exit(1); }
	if (O->int_quals) {
		/* intToPhred33 (qual.h:132-153) knows Phred and Solexa scales only */
		if (O->rd.format != BT_FMT_FASTQ) die("Error: --integer-quals is for FASTQ input");
		O->rd.qual_enc = O->rd.qual_enc == BT_QUAL_SOLEXA64 ? BT_QUAL_INT_SOLEXA : BT_QUAL_INT;
	}
	if (O->tryhard) O->pol.max_bts = INT_MAX;
	if (O->out.sam && O->suppress_set) {
		if (!O->quiet) fprintf(stderr, "Warning: Ignoring --suppress because output type is not default.\n"
		                               "         --suppress is only available for the default output type.\n");
		O->out.suppress = 0;
	}
	if (O->threads <= 0) { const unsigned hc = std::thread::hardware_concurrency(); O->threads = hc == 0 ? 1 : (hc > 32 ? 32 : (int)hc); }
	O->out.khits = O->pol.khits; O->out.mhits = O->pol.mhits; O->out.all_hits = O->pol.all_hits;
	O->out.sample_max = O->pol.sample_max;
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

/* adjustEbwtBase (ebwt.h): the prefix as given, else under $BOWTIE_INDEXES */
std::string find_index(const std::string& base)
{
	auto there = [](const std::string& b) {
		return file_exists(b + ".1.ebwt") || file_exists(b + ".1.bt2") || file_exists(b + ".1.ebwtl") || file_exists(b + ".1.bt2l");
	};
	if (there(base)) return base;
	const char* dir = getenv("BOWTIE_INDEXES");
	if (dir && *dir) {
		std::string p = std::string(dir) + (dir[strlen(dir) - 1] == '/' ? "" : "/") + base;
		if (there(p)) return p;
	}
	return base;
}

double now_s() { struct timeval tv; gettimeofday(&tv, nullptr); return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec; }
/* BT_CLI_TIMELINE=1: when each stage took up and let go of each batch, printed (seconds since the start) at the end of
 * the run -- where the pipeline of reader, searcher and writer waits */
struct Timeline {
	bool on = getenv("BT_CLI_TIMELINE") != nullptr;
	double t0 = now_s();
	std::mutex m;
	struct Ev { double t; const char* what; uint64_t seq; };
	std::vector<Ev> evs;
	void mark(const char* what, uint64_t seq) { if (!on) return; const double t = now_s() - t0; std::lock_guard<std::mutex> l(m); evs.push_back({t, what, seq}); }
	void print() { if (!on) return; for (const Ev& e : evs) fprintf(stderr, "[timeline] %9.4f  batch %-3llu %s\n", e.t, (unsigned long long)e.seq, e.what); }
};
static Timeline g_tl;
/* A piece of formatted text to the output.  stdio copies what it is given through the stream's buffer unless the piece is a
 * multiple of it: megabytes of memcpy() on the one thread that writes, most of the 0.3 s a batch of 12 M reads took to "write"
 * to /dev/null (round 6's timeline).  Large pieces go to the descriptor themselves, behind whatever the stream still holds. */
void put_text(FILE* f, const char* p, size_t n)
{
	if (n < ((size_t)1u << 16)) { fwrite(p, 1, n, f); return; }
	if (fflush(f) != 0) return;                                  /* the stream's error indicator is set: seen at the end of the run */
	const int fd = fileno(f);
	while (n) {
		const ssize_t w = write(fd, p, n);
		if (w < 0) { if (errno == EINTR) continue; fwrite(p, 1, n, f); return; }   /* let stdio meet the error and keep it */
		p += w; n -= (size_t)w;
	}
}
void print_timer(const char* msg, double secs)
{
	/* Timer::write (timer.h): hh:mm:ss */
	const long s = (long)secs;
	fprintf(stderr, "%s%02ld:%02ld:%02ld\n", msg, s / 3600, (s / 60) % 60, s % 60);
}

/* The formatter's threads, started once.  Until round 6's call 13 the writer started its threads anew for every batch: on the GPU
 * box's host (256 hardware threads) starting and joining a thread is ~0.8 ms -- its stack is mapped, and unmapped again with a
 * TLB shoot-down across the machine -- so that 128 threads were 0.1 s per batch before any of them formatted a read
 * (scripts/r6/fmt_probe.cpp: 12.6 M reads in 0.039 s on 32 fresh threads, 0.102 s on 128, 0.199 s on 256). */
class WorkPool {
public:
	explicit WorkPool(int n) { for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); }); }
	~WorkPool() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_work_.notify_all(); for (auto& t : th_) t.join(); }
	/* fn(i) for i in [0, count), handed out in order to whichever thread is free; returns at once.  wait() before the next start(). */
	void start(size_t count, std::function<void(size_t)> fn)
	{
		std::lock_guard<std::mutex> l(m_);
		fn_ = std::move(fn); count_ = count; next_.store(0); active_ = th_.size(); gen_++;
		cv_work_.notify_all();
	}
	void wait() { std::unique_lock<std::mutex> l(m_); cv_done_.wait(l, [&] { return active_ == 0; }); }
private:
	void loop()
	{
		uint64_t seen = 0;
		for (;;) {
			std::unique_lock<std::mutex> l(m_);
			cv_work_.wait(l, [&] { return stop_ || gen_ != seen; });
			if (stop_) return;
			seen = gen_;
			l.unlock();
			for (;;) { const size_t i = next_.fetch_add(1); if (i >= count_) break; fn_(i); }
			l.lock();
			if (--active_ == 0) cv_done_.notify_all();
		}
	}
	std::mutex m_; std::condition_variable cv_work_, cv_done_;
	std::vector<std::thread> th_;
	std::function<void(size_t)> fn_;
	std::atomic<size_t> next_{0};
	size_t count_ = 0, active_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};

/* ---- one batch travelling through the stages ------------------------------------------------ */
struct Job {
	bt_read_batch rb;                        /* view into `store` */
	std::unique_ptr<BtHostBatch> store;
	bt_read_batch rb2;                       /* paired-end: the second mates */
	std::unique_ptr<BtHostBatch> store2;
	uint32_t hit_cap = 1;
	PinVec<bt_hit> hits;
	PinVec<uint32_t> n_hits;
	PinVec<uint8_t> status;
	PinVec<uint16_t> mm_pool;
	uint32_t mm_used = 0;
	bool prepared = false;                   /* search_prepare has sized the four arrays for this batch */
	bt_hit_batch hb;                         /* view into the four arrays above (what the search fills) */
	/* reads whose hits did not fit the uniform slots: searched again alone with room for all */
	struct Wide { uint32_t read; uint32_t hit_cap; std::vector<bt_hit> hits; std::vector<uint16_t> pool; uint32_t n_hits; uint8_t status; };
	std::vector<Wide> wide;
	bool last = false;
	uint64_t seq = 0;                        /* position in the input: the writer restores this order */
	std::string error;
	/* a --12 batch: its unpaired records as a job of their own (searched by the unpaired stateful aligner), and how the
	 * input interleaved pairs (1) and unpaired reads (0) */
	std::unique_ptr<Job> unp;
	std::vector<uint8_t> order;
};

template <class T> class Chan {
public:
	explicit Chan(size_t cap) : cap_(cap) {}
	void put(T v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
	T take() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); T v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return v; }
	bool try_take(T* v) { std::unique_lock<std::mutex> l(m_); if (q_.empty()) return false; *v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return true; }
	bool try_put(T& v) { std::unique_lock<std::mutex> l(m_); if (q_.size() >= cap_) return false; q_.push_back(std::move(v)); cv_.notify_all(); return true; }
private:
	std::mutex m_; std::condition_variable cv_; std::deque<T> q_; size_t cap_;
};


/* one read of a batch as a batch of its own */
bt_read_batch one_read(const bt_read_batch& rb, uint32_t i)
{
	bt_read_batch one = rb;
	one.n_reads = 1;
	one.seq = rb.seq + (size_t)i * rb.stride; one.qual = rb.qual + (size_t)i * rb.stride;
	one.len = rb.len + i; one.seed = rb.seed + i;
	return one;
}

/* Search one batch.  Hit slots are uniform per read (bt_hit_batch); reads that report more than the
 * first pass had slots for (-a, large -k) or whose mismatch lists outgrew the pool are searched
 * again, together, with as many slots as the hungriest of them needs.  "" = ok. */
/* pairs: slots per pair = 2 x hits wanted; pairs that have more (-a) are searched again with room for all */
std::string search_job_pairs(bt_ctx* ctx, const Options& O, Job* j)
{
	const uint32_t n = j->rb.n_reads;
	const bool all = O.pol.all_hits != 0;
	j->hit_cap = all ? 16u : 2u * (O.pol.khits > 32u ? 32u : O.pol.khits);
	/* -M: a pair over the ceiling keeps its first mhits alignments, one of which is printed (at most 64 are kept) */
	if (O.pol.sample_max && !all) { const uint32_t w = 2u * (O.pol.mhits > 64u ? 64u : O.pol.mhits); if (w > j->hit_cap) j->hit_cap = w; }
	j->hits.assign((size_t)n * j->hit_cap, bt_hit());
	j->n_hits.assign(n, 0); j->status.assign(n, 0);
	j->mm_pool.resize((size_t)n * j->hit_cap * 6u + 1024u);
	bt_hit_batch hb = { j->hit_cap, j->hits.data(), j->n_hits.data(), j->status.data(), j->mm_pool.data(), (uint32_t)j->mm_pool.size(), 0 };
	int rc = bt_align_pairs(ctx, &j->rb, &j->rb2, &hb, nullptr);
	j->mm_used = hb.mm_pool_used;
	if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return std::string("Error: search failed: ") + bt_strerror(rc);
	/* pairs with more alignments than the uniform slots hold (-a, large -k on repetitive references), or whose
	 * mismatch lists outgrew the pool: those pairs alone are searched again with room for all, as search_finish
	 * does for unpaired reads */
	const uint32_t maxv = O.pol.mhits == 0xffffffffu ? 0xffffffffu : O.pol.mhits * 2u;
	std::vector<uint32_t> redo, need;
	for (uint32_t i = 0; i < n; i++) {
		if (j->status[i] & BT_ST_OVERFLOW) return "Error: a read exceeded the search scratch space";
		const uint32_t tot = j->n_hits[i];
		if (tot > maxv && !O.pol.sample_max) continue;               /* nothing of it is printed */
		const uint32_t want = tot > maxv ? maxv : (all ? tot : (tot < 2u * O.pol.khits ? tot : 2u * O.pol.khits));
		if (want > j->hit_cap || (j->status[i] & BT_ST_MMPOOL)) { redo.push_back(i); need.push_back(((want > j->hit_cap ? want : j->hit_cap) + 1u) & ~1u); }
	}
	size_t at = 0;
	while (at < redo.size()) {
		/* a group whose slot matrix stays under 16 M hits; a single pair may exceed that on its own */
		size_t end = at; uint32_t cap = 0; uint32_t maxlen = 6;
		while (end < redo.size()) {
			const uint32_t c2 = need[end] > cap ? need[end] : cap;
			if (end > at && (uint64_t)c2 * (end - at + 1) > (16ull << 20)) break;
			cap = c2;
			const uint32_t i = redo[end];
			if (j->rb.len[i] > maxlen) maxlen = j->rb.len[i];
			if (j->rb2.len[i] > maxlen) maxlen = j->rb2.len[i];
			end++;
		}
		if ((uint64_t)cap > (256ull << 20)) return "Error: too many alignments for one pair (more than 128 M)";
		const uint32_t m = (uint32_t)(end - at);
		BtHostBatch s1, s2;
		s1.reset(m, j->rb.stride); s2.reset(m, j->rb2.stride);
		for (uint32_t k = 0; k < m; k++) {
			const uint32_t i = redo[at + k];
			memcpy(s1.seq + (size_t)k * s1.stride, j->rb.seq + (size_t)i * j->rb.stride, j->rb.stride);
			memcpy(s1.qual + (size_t)k * s1.stride, j->rb.qual + (size_t)i * j->rb.stride, j->rb.stride);
			s1.len[k] = j->rb.len[i]; s1.seed[k] = j->rb.seed[i];
			memcpy(s2.seq + (size_t)k * s2.stride, j->rb2.seq + (size_t)i * j->rb2.stride, j->rb2.stride);
			memcpy(s2.qual + (size_t)k * s2.stride, j->rb2.qual + (size_t)i * j->rb2.stride, j->rb2.stride);
			s2.len[k] = j->rb2.len[i]; s2.seed[k] = j->rb2.seed[i];
		}
		const bt_read_batch r1 = s1.view(), r2 = s2.view();
		std::vector<bt_hit> sh((size_t)m * cap);
		std::vector<uint32_t> snh(m); std::vector<uint8_t> sst(m);
		uint64_t pool_n = (uint64_t)m * cap * maxlen + 1024u;          /* a full-length mismatch list per alignment */
		if (pool_n > 0xfffffff0ull) pool_n = 0xfffffff0ull;
		std::vector<uint16_t> sp((size_t)pool_n);
		bt_hit_batch shb = { cap, sh.data(), snh.data(), sst.data(), sp.data(), (uint32_t)sp.size(), 0 };
		rc = bt_align_pairs(ctx, &r1, &r2, &shb, nullptr);
		if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return std::string("Error: search failed: ") + bt_strerror(rc);
		for (uint32_t k = 0; k < m; k++) {
			if (sst[k] & (BT_ST_OVERFLOW | BT_ST_MMPOOL)) return "Error: a read exceeded the search scratch space";
			Job::Wide w;
			w.read = redo[at + k]; w.n_hits = snh[k]; w.status = sst[k];
			const uint32_t tot = snh[k];
			uint32_t keep = tot > maxv ? maxv : (all ? tot : (tot < 2u * O.pol.khits ? tot : 2u * O.pol.khits));
			if (keep > cap) keep = cap;
			w.hit_cap = keep ? keep : 2u;
			w.hits.assign(sh.begin() + (size_t)k * cap, sh.begin() + (size_t)k * cap + w.hit_cap);
			for (uint32_t h = 0; h < keep; h++) {              /* the pair's mismatch lists, re-based */
				bt_hit& x = w.hits[h];
				const uint32_t off = (uint32_t)w.pool.size();
				w.pool.insert(w.pool.end(), sp.begin() + x.mm_off, sp.begin() + x.mm_off + x.nmm);
				x.mm_off = off;
			}
			if (w.pool.empty()) w.pool.push_back(0);
			j->wide.push_back(std::move(w));
		}
		at = end;
	}
	return "";
}

/* size a job's result arrays for the first pass */
void search_prepare(const Options& O, Job* j)
{
	if (j->prepared) return;
	j->prepared = true;
	const uint32_t n = j->rb.n_reads;
	const bool all = O.pol.all_hits != 0;
	j->hit_cap = all ? 16u : (O.pol.khits > 64u ? 64u : O.pol.khits);
	/* -M: a read over the ceiling keeps its first mhits hits, one of which is printed */
	if (O.pol.sample_max && !all && O.pol.mhits > j->hit_cap) j->hit_cap = O.pol.mhits > 64u ? 64u : O.pol.mhits;
	j->hits.resize((size_t)n * j->hit_cap);
	j->n_hits.assign(n, 0); j->status.assign(n, 0);
	j->mm_pool.resize((size_t)n * j->hit_cap * 6u + 1024u);
	j->hb = bt_hit_batch{ j->hit_cap, j->hits.data(), j->n_hits.data(), j->status.data(), j->mm_pool.data(), (uint32_t)j->mm_pool.size(), 0 };
}

std::string search_finish(bt_ctx* ctx, const Options& O, Job* j, int rc, bool streamed_first_pass = false);

std::string search_job(bt_ctx* ctx, const Options& O, Job* j)
{
	if (O.paired) return search_job_pairs(ctx, O, j);
	search_prepare(O, j);
	const int rc = bt_align_batch(ctx, &j->rb, &j->hb, nullptr);
	return search_finish(ctx, O, j, rc);
}

/* the first pass is back (rc = what bt_align_batch said, or worked out from the status bytes after a streamed
 * search): errors the reference stops at, and the second pass for reads with more hits than slots.  ctx: a context
 * with nothing in flight. */
std::string search_finish(bt_ctx* ctx, const Options& O, Job* j, int rc, bool streamed_first_pass)
{
	const uint32_t n = j->rb.n_reads;
	const bool all = O.pol.all_hits != 0;
	j->mm_used = j->hb.mm_pool_used;
	if (rc == BT_ERR_READ_SHORT) {
		/* the reference stops at the first such read (search_1mm_phase1.c:12-15, search_23mm_phase1.c:13-20) */
		for (uint32_t i = 0; i < n; i++) if (j->status[i] & BT_ST_TOOSHORT) {
			if (O.pol.mms == 1) return "Error: Reads must be at least 2 characters long in 1-mismatch mode";
			const std::string nm(j->store->names.data() + j->store->name_off[i], (size_t)(j->store->name_off[i + 1] - j->store->name_off[i]));
			return "Error: Read (" + nm + ") is less than " + (j->rb.len[i] < 3 ? "3" : "4") + " characters long";
		}
		return "Error: read too short for the alignment mode";
	}
	if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return std::string("Error: search failed: ") + bt_strerror(rc);

	std::vector<uint32_t> redo, need;
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t tot = j->n_hits[i];
		if (j->status[i] & BT_ST_OVERFLOW) {
			/* a streamed search leaves reads that outgrew their scratch flagged: bt_align_batch below runs them again
			 * with worst-case arenas.  After that pass the flag is fatal. */
			if (!streamed_first_pass) return "Error: a read exceeded the search scratch space";
			redo.push_back(i); need.push_back(j->hit_cap);
			continue;
		}
		if (tot > O.pol.mhits && !O.pol.sample_max) continue;         /* nothing of it is printed */
		const uint32_t want = tot > O.pol.mhits ? O.pol.mhits : (all ? tot : (tot < O.pol.khits ? tot : O.pol.khits));
		if (want > j->hit_cap || (j->status[i] & BT_ST_MMPOOL)) { redo.push_back(i); need.push_back(want > j->hit_cap ? want : j->hit_cap); }
	}
	size_t at = 0;
	while (at < redo.size()) {
		/* a group whose slot matrix stays under 16 M hits */
		size_t end = at; uint32_t cap = 0; uint32_t maxlen = 1;
		while (end < redo.size()) {
			const uint32_t c2 = need[end] > cap ? need[end] : cap;
			if (end > at && (uint64_t)c2 * (end - at + 1) > (16ull << 20)) break;
			cap = c2;
			if (j->rb.len[redo[end]] > maxlen) maxlen = j->rb.len[redo[end]];
			end++;
		}
		const uint32_t m = (uint32_t)(end - at);
		BtHostBatch sub;
		sub.reset(m, j->rb.stride);
		for (uint32_t k = 0; k < m; k++) {
			const uint32_t i = redo[at + k];
			memcpy(sub.seq + (size_t)k * sub.stride, j->rb.seq + (size_t)i * j->rb.stride, j->rb.stride);
			memcpy(sub.qual + (size_t)k * sub.stride, j->rb.qual + (size_t)i * j->rb.stride, j->rb.stride);
			sub.len[k] = j->rb.len[i]; sub.seed[k] = j->rb.seed[i];
		}
		const bt_read_batch srb = sub.view();
		std::vector<bt_hit> sh((size_t)m * cap);
		std::vector<uint32_t> snh(m); std::vector<uint8_t> sst(m);
		uint64_t pool_n = (uint64_t)m * cap * (maxlen < 12u ? maxlen : 12u) + 1024u;
		if (pool_n > 0xfffffff0ull) pool_n = 0xfffffff0ull;
		std::vector<uint16_t> sp((size_t)pool_n);
		bt_hit_batch shb = { cap, sh.data(), snh.data(), sst.data(), sp.data(), (uint32_t)sp.size(), 0 };
		rc = bt_align_batch(ctx, &srb, &shb, nullptr);
		if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return std::string("Error: search failed: ") + bt_strerror(rc);
		for (uint32_t k = 0; k < m; k++) {
			Job::Wide w;
			w.read = redo[at + k]; w.n_hits = snh[k]; w.status = sst[k];
			if (sst[k] & BT_ST_OVERFLOW) return "Error: a read exceeded the search scratch space";
			const uint32_t tot = snh[k];
			uint32_t keep = tot > O.pol.mhits ? O.pol.mhits : (all ? tot : (tot < O.pol.khits ? tot : O.pol.khits));
			if (keep > cap) keep = cap;
			if (sst[k] & BT_ST_MMPOOL) {
				/* still short of mismatch slots: this read alone, a full-length list per hit */
				w.hit_cap = keep ? keep : 1u;
				w.hits.assign(w.hit_cap, bt_hit());
				w.pool.assign((size_t)w.hit_cap * j->rb.len[w.read] + 16u, 0);
				const bt_read_batch one = one_read(j->rb, w.read);
				bt_hit_batch hw = { w.hit_cap, w.hits.data(), &w.n_hits, &w.status, w.pool.data(), (uint32_t)w.pool.size(), 0 };
				rc = bt_align_batch(ctx, &one, &hw, nullptr);
				if (rc != BT_OK && rc != BT_ERR_OVERFLOW) return std::string("Error: search failed: ") + bt_strerror(rc);
				if (w.status & (BT_ST_MMPOOL | BT_ST_OVERFLOW)) return "Error: a read exceeded the search scratch space";
			} else {
				w.hit_cap = keep ? keep : 1u;
				w.hits.assign(sh.begin() + (size_t)k * cap, sh.begin() + (size_t)k * cap + w.hit_cap);
				for (uint32_t h = 0; h < keep; h++) {          /* the read's mismatch lists, re-based */
					bt_hit& x = w.hits[h];
					const uint32_t off = (uint32_t)w.pool.size();
					w.pool.insert(w.pool.end(), sp.begin() + x.mm_off, sp.begin() + x.mm_off + x.nmm);
					x.mm_off = off;
				}
				if (w.pool.empty()) w.pool.push_back(0);
			}
			j->wide.push_back(std::move(w));
		}
		at = end;
	}
	return "";
}

}  // namespace


/* the read stream(s) of the run: one for unpaired input, one per mate for pairs (for --12 / --interleaved both over
 * the same file, each keeping its mate) */
void open_read_streams(Options& O, BtReadStream** rs, BtReadStream** rs2)
{
	std::string open_err;
	const bool dumping = !O.dump_al.empty() || !O.dump_un.empty() || !O.dump_max.empty();
	if (dumping) O.rd.flags |= BT_READ_KEEP_RAW;
	bt_read_opts rd1 = O.rd, rd2 = O.rd;
	rd1.flags |= BT_READ_MATE1; rd2.flags |= BT_READ_MATE2;
	if (O.interleaved) { rd1.flags |= BT_READ_INTERLEAVED; rd2.flags |= BT_READ_INTERLEAVED; }
	*rs = bt_io_open(O.paired ? O.mates1.c_str() : O.reads.c_str(), O.paired ? rd1 : O.rd, &open_err);
	*rs2 = O.paired ? bt_io_open(O.mates2.c_str(), rd2, &open_err) : nullptr;
}

/* BT_CLI_INPUT_ONLY=1 (tests, no GPU needed): parse the input the way the run would and list it -- one line per read,
 * "<name>\t<length>" (pairs: both mates on the line) -- without loading an index */
int list_input(Options& O)
{
	BtReadStream* rs = nullptr; BtReadStream* rs2 = nullptr;
	open_read_streams(O, &rs, &rs2);
	const bool tabbed = O.rd.format == BT_FMT_TABBED;
	BtHostBatch b1, b2;
	std::string err;
	for (;;) {
		int r = bt_io_next(rs, O.batch_reads, O.threads, &b1, &err);
		BtHostBatch bu;
		std::vector<uint8_t> order;
		if (r == BT_OK && O.paired) {
			r = bt_io_next(rs2, O.batch_reads, O.threads, &b2, &err);
			if (r == BT_OK && !bt_io_intersect_pairs(&b1, &b2)) {
				err = b1.end_rdid < b2.end_rdid ? "Error, fewer reads in file specified with -1 than in file specified with -2"
				                                : "Error, fewer reads in file specified with -2 than in file specified with -1";
				r = BT_ERR_READS;
			}
			if (r == BT_OK && tabbed && !bt_io_split_tabbed(&b1, &b2, &bu, &order)) { err = "Error: internal: a --12 batch lost its pair flags"; r = BT_ERR_READS; }
		}
		if (r != BT_OK) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
		if (b1.n == 0 && bu.n == 0) break;
		if (order.empty()) order.assign(b1.n, 1);
		uint32_t ip = 0, iu = 0;
		for (uint8_t isp : order) {
			if (!isp) { printf("%.*s\t%u\n", (int)(bu.name_off[iu + 1] - bu.name_off[iu]), bu.names.data() + bu.name_off[iu], (unsigned)bu.len[iu]); iu++; continue; }
			const uint32_t i = ip++;
			printf("%.*s\t%u", (int)(b1.name_off[i + 1] - b1.name_off[i]), b1.names.data() + b1.name_off[i], (unsigned)b1.len[i]);
			if (O.paired) printf("\t%.*s\t%u", (int)(b2.name_off[i + 1] - b2.name_off[i]), b2.names.data() + b2.name_off[i], (unsigned)b2.len[i]);
			printf("\n");
		}
	}
	bt_io_close(rs); if (rs2) bt_io_close(rs2);
	return 0;
}

int main(int argc, char** argv)
{
	Options O;
	parse_args(argc, argv, &O);
	if (getenv("BT_CLI_INPUT_ONLY")) return list_input(O);
	const double t_all = now_s();

	/* ---- the first batch of reads is parsed while the index loads ---- */
	BtReadStream* rs = nullptr; BtReadStream* rs2 = nullptr;
	/* BT_CLI_PINNED=1: read batches live in page-locked memory from the library, so that bt_align_stream_submit's uploads
	 * are DMAs beside the running search (not the default until it has been measured on the GPU) */
	if (getenv("BT_CLI_PINNED") && atoi(getenv("BT_CLI_PINNED")) != 0) bt_io_set_allocator(bt_host_alloc, bt_host_free);          /* see `pinned` below */
	else if (bigmem::on()) bt_io_set_allocator(big_alloc, big_free);
	open_read_streams(O, &rs, &rs2);
	const bool tabbed = O.rd.format == BT_FMT_TABBED;
	const bool dumping = !O.dump_al.empty() || !O.dump_un.empty() || !O.dump_max.empty();
	const int T = O.threads;
	/* the formatter's threads: -p of them, or twice that where the host has the cores to spare (the parser's -p threads and the
	 * formatter's are busy at the same time only in the middle of a run; at its end the formatter works through the last
	 * batches alone, and that stretch is part of every run: 5 of 28 s at 192 M reads in round 6's timeline) */
	int TF = T;
	{
		const unsigned hw = std::thread::hardware_concurrency();
		const int half = (int)(hw / 2u);
		const int cap = half > T ? half : T;
		TF = 2 * T < cap ? 2 * T : cap;
		if (const char* e = getenv("BT_CLI_FORMAT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) TF = v; }
	}
	double busy_read = 0, busy_write = 0;
	std::atomic<bool> abort_run(false);
	Chan<std::unique_ptr<BtHostBatch>> spare(8);              /* read batches on their way back to the reader */
	/* ... and result arrays: a batch's four arrays (some 150 MB for 4 M reads) keep their memory from batch to batch */
	struct ResultBufs { PinVec<bt_hit> hits; PinVec<uint32_t> n_hits; PinVec<uint8_t> status; PinVec<uint16_t> mm_pool; };
	Chan<std::unique_ptr<ResultBufs>> spare_res(8);
	/* What neither channel has room for is let go on a thread of its own: unmapping a batch's 5 GB took the writer a third of
	 * a second per batch at the end of a run, where it is the only stage still working (round 6's timeline).  Once the input is
	 * exhausted nothing is kept back for the reader any more. */
	struct Trash { std::unique_ptr<Job> j; std::unique_ptr<ResultBufs> r; std::unique_ptr<BtHostBatch> s; std::vector<std::string> text; };
	Chan<std::unique_ptr<Trash>> to_reap(64);
	std::atomic<bool> input_done(false);
	const bool full_teardown = getenv("BT_CLI_TEARDOWN") && atoi(getenv("BT_CLI_TEARDOWN")) != 0;
	int n_reapers = 4;                                         /* unmapping runs beside unmapping (the address space is only read-locked meanwhile) */
	if (const char* e = getenv("BT_CLI_REAPERS")) { const int v = atoi(e); if (v >= 1 && v <= 16) n_reapers = v; }
	auto reap = [&] {
		for (;;) {
			std::unique_ptr<Trash> t = to_reap.take();
			if (!t) return;
			t.reset();
			if (input_done.load()) {
				std::unique_ptr<BtHostBatch> b; std::unique_ptr<ResultBufs> r;
				while (spare.try_take(&b)) b.reset();
				while (spare_res.try_take(&r)) r.reset();
			}
		}
	};
	auto let_go = [&](std::unique_ptr<Job>& j, std::unique_ptr<ResultBufs>& r, std::unique_ptr<BtHostBatch>& st) {
		std::unique_ptr<Trash> t(new Trash());
		t->j = std::move(j); t->r = std::move(r); t->s = std::move(st);
		to_reap.put(std::move(t));
	};
	const bool will_stream = !O.paired && !O.pol.best && !O.no_stream;      /* what `streamed` below says once the index is there */
	bool big_host = false;                                     /* 256 GB of memory and more */
	{
		const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
		big_host = pages > 0 && psz > 0 && (double)pages * (double)psz >= 256e9;
	}
	if (will_stream && !O.batch_set) {
		/* (see Options::batch_reads) */
		if (big_host) O.batch_reads = 12u << 20;
	}
	/* one batch from the input into a job: returns BT_OK or the error it left in j->error */
	auto read_job = [&](Job* j) -> int {
		if (!spare.try_take(&j->store)) j->store.reset(new BtHostBatch());
		std::string err;
		const double tb = now_s();
		int r = BT_OK;
		if (abort_run.load()) j->store->n = 0;
		else r = bt_io_next(rs, O.batch_reads, T, j->store.get(), &err);
		busy_read += now_s() - tb;
		if (r != BT_OK) j->error = err;
		j->rb = j->store->view();
		if (O.paired && r == BT_OK) {
			j->store2.reset(new BtHostBatch());
			if (abort_run.load()) j->store2->n = 0;
			else r = bt_io_next(rs2, O.batch_reads, T, j->store2.get(), &err);
			if (r != BT_OK) j->error = err;
			/* a mate record that does not parse takes its pair out of the run (pat.cpp:96-127), not just itself */
			const bool same_end = r != BT_OK || bt_io_intersect_pairs(j->store.get(), j->store2.get());
			j->rb = j->store->view();
			j->rb2 = j->store2->view();
			if (r == BT_OK && !same_end) {
				/* PatternComposer (pat.cpp:198-212) */
				j->error = j->store->end_rdid < j->store2->end_rdid ? "Error, fewer reads in file specified with -1 than in file specified with -2"
				                                                    : "Error, fewer reads in file specified with -2 than in file specified with -1";
				r = BT_ERR_READS;
			}
			if (r == BT_OK && tabbed) {
				/* a --12 batch: the unpaired records leave for a job of their own */
				std::unique_ptr<Job> u(new Job());
				u->store.reset(new BtHostBatch());
				if (!bt_io_split_tabbed(j->store.get(), j->store2.get(), u->store.get(), &j->order)) { j->error = "Error: internal: a --12 batch lost its pair flags"; r = BT_ERR_READS; }
				j->rb = j->store->view(); j->rb2 = j->store2->view();
				if (u->store->n) { u->rb = u->store->view(); j->unp = std::move(u); }
				else j->order.clear();
			}
		}
		if (r == BT_OK && will_stream && j->rb.n_reads) {
			/* the streamed searcher's thread is the one the GPU waits for: the batch's result arrays are sized here, on the
			 * reader's thread, out of recycled memory */
			const double tp = now_s();
			std::unique_ptr<ResultBufs> rbuf;
			if (spare_res.try_take(&rbuf)) { j->hits.swap(rbuf->hits); j->n_hits.swap(rbuf->n_hits); j->status.swap(rbuf->status); j->mm_pool.swap(rbuf->mm_pool); }
			search_prepare(O, j);
			busy_read += now_s() - tp;
		}
		return r;
	};
	/* the index is located before the first batch's thread starts: find_index() leaves through exit() when there is none,
	 * and no thread of ours may be inside the reader then */
	if (O.large_index) {
		/* bowtie --large-index: "force usage of a 'large' index, even if a small one is present" -- the wrapper starts
		 * bowtie-align-l, whose loader knows only <base>.*.ebwtl.  Here: the binary with 64-bit rows, its loader looking for
		 * the 64-bit files first (BT_INDEX_PREFER_LARGE, bt_host.cpp) */
		setenv("BT_INDEX_PREFER_LARGE", "1", 1);
		if (!bt_rows64()) {
			char self[PATH_MAX];
			const ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 3);
			if (n > 0) {
				self[n] = 0;
				strcat(self, "-l");
				fflush(stdout); fflush(stderr);
				execv(self, argv);
			}
			die("Error: --large-index needs the 64-bit-row build (bowtie-amd-l), which could not be started");
		}
	}
	const std::string base = find_index(O.index);
	if (!bt_rows64() && bt_index_needs_rows64(base.c_str()) == 1) {
		/* 2^32-1 rows or more: the reference's wrapper starts bowtie-align-l for such an index (bowtie:52-81); here the
		 * same sources built with 64-bit rows, next to this binary.  Decided from the index's header NOW, before the first
		 * batch's thread takes a single read from the input: what comes from a pipe cannot be read twice */
		char self[PATH_MAX];
		const ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 3);
		if (n > 0) {
			self[n] = 0;
			strcat(self, "-l");
			fflush(stdout); fflush(stderr);
			execv(self, argv);
		}
		die("Error: index \"%s\" has 2^32-1 rows or more and the 64-bit-row build (bowtie-amd-l) could not be started", O.index.c_str());
	}
	if (O.devices.empty()) O.devices.push_back(0);
	std::unique_ptr<Job> first_job(new Job());
	int first_rc = BT_OK;
	const bool pinned = getenv("BT_CLI_PINNED") && atoi(getenv("BT_CLI_PINNED")) != 0;
	std::thread prefetch([&] {
		first_rc = read_job(first_job.get());
		/* page-locking a batch's memory takes a good part of the time it takes to parse one: with BT_CLI_PINNED the batches
		 * that will be recycled for the rest of the run are made here, while the index loads, not one per batch later */
		if (pinned && first_rc == BT_OK && !O.paired && first_job->rb.n_reads == O.batch_reads)
			for (int k = 0; k < 5; k++) {
				std::unique_ptr<BtHostBatch> b(new BtHostBatch());
				b->reset(first_job->rb.n_reads, first_job->rb.stride);
				b->n = 0;
				if (!spare.try_put(b)) break;
			}
	});

	const size_t ND = O.devices.size();
	std::vector<bt_index*> idxs(ND, nullptr);
	double t0 = now_s();
	int rc = BT_OK;
	{
		/* one replica per GPU, loaded side by side */
		std::vector<int> rcs(ND, BT_OK);
		std::vector<std::thread> th;
		for (size_t d = 0; d < ND; d++) th.emplace_back([&, d] { rcs[d] = bt_index_load(base.c_str(), 1, O.offrate, O.devices[d], &idxs[d]); });
		for (auto& x : th) x.join();
		for (size_t d = 0; d < ND; d++) if (rcs[d] != BT_OK) { rc = rcs[d]; break; }
	}
	/* ---- index into HBM (above) ---- */
	if (rc != BT_OK) {
		/* said at once -- not after a first batch has been waited for (reads on a pipe may be a long time coming).  The
		 * reader's thread may be inside the input at this moment: the process leaves without running static destructors
		 * under it (nothing has been written yet) */
		/* (an index of 2^32-1 rows or more never gets here in the 32-bit build: see above, before the first read was taken) */
		if (rc == BT_ERR_ROWS64) fprintf(stderr, "Error: index \"%s\" has 2^32-1 rows or more: it needs the 64-bit-row build (bowtie-amd-l)\n", O.index.c_str());
		else if (rc == BT_ERR_IO) fprintf(stderr, "Could not locate a Bowtie index corresponding to basename \"%s\"\n", O.index.c_str());
		else fprintf(stderr, "Error: could not load index \"%s\": %s\n", O.index.c_str(), bt_strerror(rc));
		fflush(nullptr);
		_exit(1);
	}
	prefetch.join();
	bt_index* idx = idxs[0];
	if (O.timing) print_timer("Time loading forward and mirror index: ", now_s() - t0);
	g_tl.mark("index loaded", 0);
	if (O.paired) {
		const double tr = now_s();
		for (size_t d = 0; d < ND; d++) {
			rc = bt_index_load_reference(idxs[d]);
			if (rc != BT_OK) die("Could not open reference-string index file %s.3.ebwt / .4.ebwt for reading: %s", base.c_str(), bt_strerror(rc));
		}
		if (O.timing) print_timer("Time loading reference: ", now_s() - tr);
	}
	bt_index_info info;
	bt_index_info_get(idx, &info);
	if (!getenv("BT_LOCUS")) {
		/* The locus image (include/bowtie_amd.h: locus mode) is derived on the GPU when the first context is created --
		 * 3.9 s for a 2.86 Gbp genome -- and makes the search 1.2-1.7 times as fast; on that genome it pays from about a
		 * hundred million reads on.  The input's size decides (a file of FASTQ is ~2.1 bytes per base; compressed, a fourth
		 * of that); input of unknown size (standard input) gets the image.  BT_LOCUS=0 / 1 overrides. */
		uint64_t bytes = 0; bool known = true;
		for (const std::string* lst : {&O.reads, &O.mates1, &O.mates2, &O.tab12, &O.ileaved})
			for (const std::string& f : split_commas(*lst)) {
				struct stat st;
				if (f == "-" || stat(f.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { known = false; continue; }
				const bool gz = f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0;
				bytes += (uint64_t)st.st_size * (gz ? 4u : 1u);
			}
		if (known && O.rd.format != BT_FMT_CMDLINE && bytes < 8ull * info.len) setenv("BT_LOCUS", "0", 1);
		else if (O.rd.format == BT_FMT_CMDLINE) setenv("BT_LOCUS", "0", 1);        /* -c: a handful of reads */
	}
	BtRefNames refs;
	for (uint32_t i = 0; i < info.n_pat; i++) { const char* nm = bt_index_refname(idx, i); refs.names.emplace_back(nm ? nm : ""); refs.lens.push_back(bt_index_reflen(idx, i)); }
	/* `--inflight` whole batches are searched side by side, each on its own context and stream.  --stream
	 * (opt-in: unpaired, phase-program engine) instead keeps one context per GPU fed through
	 * bt_align_stream_* with the reads a batch leaves running carried into the next ones (bt_ctx_set_carry). */
	const bool streamed = !O.paired && !O.pol.best && !O.no_stream;      /* --stream is the default (round 3: GPU-verified) */
	/* streamed: BT_CLI_STREAMS contexts per GPU, each fed its own run of batches with its own carry-over chain -- a launch
	 * ends with a stretch in which a few wavefronts finish what has been carried long enough, and launches of one context
	 * follow each other on one stream: with a second context the other's launch fills the machine meanwhile (round 6's
	 * timeline of a 192 M-read run: one batch per 0.7-1.1 s GPU-side where the kernel needs 0.54 s) */
	int n_streams = 1;
	if (const char* sv = getenv("BT_CLI_STREAMS")) { n_streams = atoi(sv); if (n_streams < 1) n_streams = 1; if (n_streams > 4) n_streams = 4; }
	std::vector<bt_ctx*> ctxs((size_t)(streamed ? n_streams : O.inflight) * ND, nullptr);          /* searcher g works on GPU g % ND */
	std::vector<bt_ctx*> redo_ctxs(streamed ? ctxs.size() : 0, nullptr);
	/* --12 input: the file's unpaired records run the stateful unpaired aligner (ebwt_search.cpp:3001-3002,
	 * MixedMultiAligner) -- the best-first engine without the pair machinery, on a context of its own */
	Options OU = O;
	OU.paired = false; OU.pol.pe_v1 = 0; OU.pol.best = 1;
	if (!O.maxbts_set) OU.pol.max_bts = 800;
	std::vector<bt_ctx*> unp_ctxs(O.rd.format == BT_FMT_TABBED && O.paired ? ctxs.size() : 0, nullptr);
	size_t fl_lim = 13;                                       /* batches a streamed searcher keeps in flight */
	int n_ticks = 14;                                         /* at the end of the input: as many as a read may ride along, and two */
	for (size_t g = 0; g < ctxs.size(); g++) {
		rc = bt_ctx_create(idxs[g % ND], &O.pol, nullptr, &ctxs[g]);
		if (rc != BT_OK) die("Error: bad alignment options: %s", bt_strerror(rc));
		if (!unp_ctxs.empty()) {
			rc = bt_ctx_create(idxs[g % ND], &OU.pol, nullptr, &unp_ctxs[g]);
			if (rc != BT_OK) die("Error: bad alignment options: %s", bt_strerror(rc));
		}
		if (streamed) {
			/* How long a read may ride along, and how many batches are in flight for it.  A batch is complete when its hardest
			 * read is: at 12 M reads per batch that is fourteen or fifteen launches later (some ten seconds of sharing a
			 * wavefront with 63 others).  With twelve launches to ride (the library's limit until round 6: a ring of 16 batches)
			 * every launch from the thirteenth on closed with ~0.5 s of the machine waiting for the batch twelve back
			 * (profiles/r6/call9_cli_192m_timeline.txt).  On a host with the memory for it (a batch in flight is 5 GB there, and
			 * 2.2 GB of HBM) reads ride up to 22 launches and 24 batches may be in flight -- few runs get there: batches leave
			 * as they complete; elsewhere round 5's 12 and 13.  BT_CLI_CARRY / BT_CLI_INFLIGHT set them (diagnostics). */
			const char* cv = getenv("BT_CLI_CARRY");
			int cage = cv && *cv ? atoi(cv) : (big_host && n_streams == 1 ? 22 : 12 / n_streams);  /* (of n contexts each sees every n-th batch: the same time to ride along) */
			if (cage > 60) cage = 60;
			if (cage < 1) cage = 1;
			fl_lim = (size_t)cage + (big_host || n_streams > 1 ? 2u : 1u);
			/* at cage + 1 the batch behind the oldest one's last launch is submitted only when that launch has ended; at cage + 2
			 * it is enqueued behind it.  Fewer than cage + 1 would wait for a batch that cannot complete yet. */
			if (const char* fv = getenv("BT_CLI_INFLIGHT")) { const int v = atoi(fv); if (v >= cage + 1 && v <= 62) fl_lim = (size_t)v; }
			n_ticks = cage + 2;
			if (bt_ctx_set_carry(ctxs[g], cage) != BT_OK) die("Error: bt_ctx_set_carry failed");
			rc = bt_ctx_create(idxs[g % ND], &O.pol, nullptr, &redo_ctxs[g]);
			if (rc != BT_OK) die("Error: bad alignment options: %s", bt_strerror(rc));
		}
	}

	/* ---- output ---- */
	FILE* fout = stdout;
	if (!O.hits_file.empty()) { fout = fopen(O.hits_file.c_str(), "wb"); if (!fout) die("Error: Could not open alignment output file %s", O.hits_file.c_str()); }
	static char obuf[1 << 22];
	setvbuf(fout, obuf, _IOFBF, sizeof(obuf));
	if (O.out.sam && !O.sam_nohead) {
		std::string rg;
		for (size_t i = 0; i < O.rg_fields.size(); i++) { if (i) rg.push_back('\t'); rg.append(O.rg_fields[i]); }
		std::string h;
		bt_io_sam_header(refs, O.out, O.cmdline.c_str(), rg.empty() ? nullptr : rg.c_str(), &h);
		fwrite(h.data(), 1, h.size(), fout);
	}

	/* ---- stage 1: reader ---- */
	const int G = (int)ctxs.size();
	/* the searcher never waits for the writer: a batch's results are a few hundred MB of host memory, and a searcher held up
	 * here stops feeding the GPU (round 4's timeline of a 64 M-read run: 1.8 s of every 10 with nothing enqueued) */
	/* the results queue: deep enough that the searcher does not wait for the writer while the in-flight batches come back in a
	 * burst (up to BT_BATCH_RING - 2 = 14 of them), not so deep that a slow writer -- SAM formatting, a slow file system, the
	 * --al/--un dumps -- lets batches of gigabytes pile up on the host without bound (round 4 had G + 64) */
	Chan<std::unique_ptr<Job>> to_gpu(2), to_out((size_t)G + 8);
	std::vector<double> busy_gpu((size_t)G, 0.0);
	std::vector<std::thread> reapers;
	for (int i = 0; i < n_reapers; i++) reapers.emplace_back(reap);
	std::thread reader([&] {
		uint64_t seq = 0;
		for (;;) {
			std::unique_ptr<Job> j;
			int r;
			if (first_job) { j = std::move(first_job); r = first_rc; g_tl.mark("read: parsed while the index was loading", seq); }
			else { g_tl.mark("read: begin", seq); j.reset(new Job()); r = read_job(j.get()); g_tl.mark("read: parsed", seq); }
			j->seq = seq++;
			if (r != BT_OK || (j->rb.n_reads == 0 && !j->unp)) {
				/* the end (or an input error, reported in its place in the order): one marker per searcher */
				j->last = true;
				input_done.store(true);
				if (!full_teardown) g_leave_pinned.store(true);
				const uint64_t sq = j->seq;
				to_gpu.put(std::move(j));
				for (int g = 1; g < G; g++) { std::unique_ptr<Job> e(new Job()); e->last = true; e->seq = sq + (uint64_t)g; to_gpu.put(std::move(e)); }
				/* the input is through: its window (a batch's raw text: 2.9 GB) goes now, beside the search of the last batches */
				bt_io_close(rs); rs = nullptr;
				if (rs2) { bt_io_close(rs2); rs2 = nullptr; }
				return;
			}
			const uint64_t sq = j->seq;
			to_gpu.put(std::move(j));
			g_tl.mark("read: handed to the searcher's queue", sq);
		}
	});

	/* ---- stage 3: writer ---- */
	bt_out_tally tally = {0, 0, 0, 0, 0, 0};
	std::string fatal;
	FILE *f_al = nullptr, *f_un = nullptr, *f_max = nullptr;
	FILE *f_al2 = nullptr, *f_un2 = nullptr, *f_max2 = nullptr;      /* pairs: the second mates' files */
	std::thread writer([&] {
		std::vector<std::string> parts;                      /* formatted text, one buffer per piece of a batch */
		std::unique_ptr<WorkPool> pool;                      /* the formatter's threads */
		auto write_batches = [&] {
		std::vector<std::unique_ptr<Job>> held;              /* finished out of turn */
		uint64_t next_seq = 0; int lasts = 0;
		for (;;) {
			std::unique_ptr<Job> j;
			for (size_t h = 0; h < held.size(); h++) if (held[h]->seq == next_seq) { j = std::move(held[h]); held.erase(held.begin() + (long)h); break; }
			if (!j) {
				if (lasts == G && held.empty()) return;
				std::unique_ptr<Job> in = to_out.take();
				if (in->seq != next_seq) { held.push_back(std::move(in)); continue; }
				j = std::move(in);
			}
			next_seq++;
			if (!j->error.empty() && fatal.empty()) { fatal = j->error; abort_run.store(true); }
			if (j->last) { if (++lasts == G && held.empty()) return; continue; }
			if (!fatal.empty()) continue;
			const double tb = now_s();
			if (j->unp) {
				/* a --12 batch with unpaired records: pairs and unpaired reads were searched as two jobs; the output
				 * follows the input, run by run of the same kind */
				Job* const JP = j.get(); Job* const JU = j->unp.get();
				auto name_of = [](const BtHostBatch& b, uint32_t i) { return std::string(b.names.data() + b.name_off[i], (size_t)(b.name_off[i + 1] - b.name_off[i])); };
				auto format_run = [&](Job* J, bool pairs, uint32_t lo, uint32_t hi, std::string* text) {
					bt_hit_batch hb = { J->hit_cap, J->hits.data(), J->n_hits.data(), J->status.data(), J->mm_pool.data(), (uint32_t)J->mm_pool.size(), J->mm_used };
					const char* names = J->store->names.data(); const uint64_t* noff = J->store->name_off.data();
					uint32_t at = lo;
					while (at < hi) {
						/* the next read of the run that was searched again with wider slots, if any */
						const Job::Wide* w = nullptr;
						for (auto& x : J->wide) if (x.read >= at && x.read < hi && (!w || x.read < w->read)) w = &x;
						const uint32_t upto = w ? w->read : hi;
						bt_out_tally tl = {0, 0, 0, 0, 0, 0};
						if (upto > at) {
							if (pairs) bt_io_format_pairs(J->rb, names, noff, J->rb2, J->store2->names.data(), J->store2->name_off.data(), hb, refs, O.out, at, upto, text, &tl);
							else bt_io_format(J->rb, names, noff, hb, refs, O.out, at, upto, text, &tl);
						}
						if (w) {
							Job::Wide& ww = const_cast<Job::Wide&>(*w);
							bt_hit_batch hw = { ww.hit_cap, ww.hits.data(), &ww.n_hits, &ww.status, ww.pool.data(), (uint32_t)ww.pool.size(), 0 };
							const bt_read_batch one1 = one_read(J->rb, ww.read);
							const uint64_t off1[2] = { noff[ww.read], noff[ww.read + 1] };
							if (pairs) {
								const bt_read_batch one2 = one_read(J->rb2, ww.read);
								const uint64_t* noff2 = J->store2->name_off.data();
								const uint64_t off2[2] = { noff2[ww.read], noff2[ww.read + 1] };
								bt_io_format_pairs(one1, names, off1, one2, J->store2->names.data(), off2, hw, refs, O.out, 0, 1, text, &tl);
							} else bt_io_format(one1, names, off1, hw, refs, O.out, 0, 1, text, &tl);
						}
						tally.aligned += tl.aligned; tally.unaligned += tl.unaligned; tally.maxed += tl.maxed; tally.reported += tl.reported;
						tally.sample_max |= tl.sample_max; tally.reported_paired += tl.reported_paired;
						at = w ? w->read + 1 : hi;
					}
				};
				auto total_hits = [](Job* J, uint32_t i) { for (auto& x : J->wide) if (x.read == i) return x.n_hits; return J->n_hits[i]; };
				std::string text;
				uint32_t ip = 0, iu = 0;
				size_t at = 0;
				while (at < j->order.size() && fatal.empty()) {
					const bool pairs = j->order[at] != 0;
					size_t end = at;
					while (end < j->order.size() && (j->order[end] != 0) == pairs) end++;
					const uint32_t cnt = (uint32_t)(end - at);
					Job* J = pairs ? JP : JU;
					uint32_t& cur = pairs ? ip : iu;
					if (!O.quiet) for (uint32_t i = cur; i < cur + cnt; i++) {
						if (pairs && (J->rb.len[i] < 4u || J->rb2.len[i] < 4u)) fprintf(stderr, "Warning: Skipping pair %s because a mate is less than 4 characters long\n", name_of(*J->store, i).c_str());
						if (!pairs && J->rb.len[i] < 4u) fprintf(stderr, "Warning: Skipping read %s because it is less than 4 characters long\n", name_of(*J->store, i).c_str());
					}
					format_run(J, pairs, cur, cur + cnt, &text);
					if (dumping) {
						/* one record per read or pair, as it stood in the file (HitSink::dumpAlign / dumpUnal / dumpMaxed with
						 * onePairFile_, hit.h:385-488); the -m ceiling counts mate alignments for pairs */
						const uint32_t ceiling = pairs ? (O.pol.mhits == 0xffffffffu ? 0xffffffffu : O.pol.mhits * 2u) : O.pol.mhits;
						for (uint32_t i = cur; i < cur + cnt; i++) {
							const uint32_t tot = total_hits(J, i);
							FILE** f; const std::string* nm;
							if (tot == 0) { f = &f_un; nm = &O.dump_un; }
							else if (tot > ceiling) { if (!O.dump_max.empty()) { f = &f_max; nm = &O.dump_max; } else { f = &f_un; nm = &O.dump_un; } }
							else { f = &f_al; nm = &O.dump_al; }
							if (nm->empty()) continue;
							if (!*f) { *f = fopen(nm->c_str(), "wb"); if (!*f) { fatal = "Error: could not open read dump file " + *nm; abort_run.store(true); break; } }
							const BtHostBatch& sb = *J->store;
							fwrite(sb.raw.data() + sb.raw_off[i], 1, (size_t)(sb.raw_off[i + 1] - sb.raw_off[i]), *f);
						}
					}
					cur += cnt;
					at = end;
				}
				put_text(fout, text.data(), text.size());
				busy_write += now_s() - tb;
				j->wide.clear();
				{
					std::unique_ptr<ResultBufs> none;
					std::unique_ptr<BtHostBatch> st = std::move(j->store);
					if (!input_done.load() && spare.try_put(st)) st.reset();
					let_go(j, none, st);
				}
				continue;
			}
			const uint32_t n = j->rb.n_reads;
			g_tl.mark("write: begin", j->seq);
			if (!O.quiet && O.paired) {
				/* PairedBWAlignerV2::setQuery (aligner.h:1579-1588) */
				for (uint32_t i = 0; i < n; i++) if (j->rb.len[i] < 4u || j->rb2.len[i] < 4u) {
					const std::string nm(j->store->names.data() + j->store->name_off[i], (size_t)(j->store->name_off[i + 1] - j->store->name_off[i]));
					fprintf(stderr, "Warning: Skipping pair %s because a mate is less than 4 characters long\n", nm.c_str());
				}
			} else if (!O.quiet && (O.pol.mode == BT_MODE_N || O.pol.best)) {
				/* search_seeded_phase1.c:17-20 / UnpairedAlignerV2::setQuery (aligner.h:440-444) */
				for (uint32_t i = 0; i < n; i++) if (j->rb.len[i] < 4u) {
					const std::string nm(j->store->names.data() + j->store->name_off[i], (size_t)(j->store->name_off[i + 1] - j->store->name_off[i]));
					if (O.pol.best) fprintf(stderr, "Warning: Skipping read %s because it is less than 4 characters long\n", nm.c_str());
					else fprintf(stderr, "Warning: Skipping read (%s) because it is less than 4 characters long\n", nm.c_str());
				}
			}
			bt_hit_batch hb = { j->hit_cap, j->hits.data(), j->n_hits.data(), j->status.data(), j->mm_pool.data(), (uint32_t)j->mm_pool.size(), j->mm_used };
			const char* names = j->store->names.data();
			const uint64_t* noff = j->store->name_off.data();
			/* segments between the reads that were searched again with wider slots */
			std::vector<uint32_t> cuts; cuts.push_back(0);
			for (auto& w : j->wide) { cuts.push_back(w.read); cuts.push_back(w.read + 1); }
			cuts.push_back(n);
			std::vector<bt_out_tally> tl;
			struct Seg { uint32_t lo, hi; int wide; };
			std::vector<Seg> segs;
			for (size_t c = 0; c + 1 < cuts.size(); c++) {
				const bool is_wide = (c & 1u) != 0;
				uint32_t lo = cuts[c], hi = cuts[c + 1];
				if (lo >= hi) continue;
				if (is_wide) { segs.push_back({lo, hi, (int)(c / 2)}); continue; }
				/* split plain segments across the threads: several pieces each, so that the first ones are on their way to
				 * the file while the rest are still being formatted */
				uint32_t pieces = (hi - lo) >= 8192 ? (uint32_t)TF * 4u : 1u;
				if (pieces > 1u && pieces > (hi - lo) / 4096u) pieces = (hi - lo) / 4096u;      /* >= 2: hi - lo >= 8192 */
				for (uint32_t p = 0; p < pieces; p++)
					segs.push_back({lo + (uint32_t)((uint64_t)(hi - lo) * p / pieces), lo + (uint32_t)((uint64_t)(hi - lo) * (p + 1) / pieces), -1});
			}
			/* the text buffers keep their memory from batch to batch */
			if (parts.size() < segs.size()) parts.resize(segs.size());
			tl.assign(segs.size(), bt_out_tally{0, 0, 0, 0, 0, 0});
			{
				const size_t per_read = (O.paired ? 2u : 1u) * (2u * (size_t)j->rb.stride + 96u) + (size_t)(noff[n] / (n ? n : 1u));
				for (size_t si = 0; si < segs.size(); si++) {
					parts[si].clear();
					const size_t want = (size_t)(segs[si].hi - segs[si].lo) * per_read;
					if (segs[si].wide < 0 && parts[si].capacity() < want) parts[si].reserve(want);
				}
			}
			/* A piece's text and tally are the formatting thread's own while it works: the strings' headers (two to a cache line,
			 * their length written twice per record) and the tallies (48 bytes each, counted up per read) of neighbouring pieces
			 * were being written by different threads at once -- 128 threads took 0.3 s over a batch that is 0.05 s of work each */
			auto run_into = [&](size_t si, std::string& text, bt_out_tally& t) {
				const Seg& sg = segs[si];
				if (O.paired) {
					if (sg.wide < 0) {
						bt_io_format_pairs(j->rb, names, noff, j->rb2, j->store2->names.data(), j->store2->name_off.data(), hb, refs, O.out,
						                   sg.lo, sg.hi, &text, &t);
						return;
					}
					/* a pair that was searched again with room for all its alignments: a one-pair view */
					Job::Wide& w = j->wide[(size_t)sg.wide];
					const bt_read_batch one1 = one_read(j->rb, w.read), one2 = one_read(j->rb2, w.read);
					bt_hit_batch hw = { w.hit_cap, w.hits.data(), &w.n_hits, &w.status, w.pool.data(), (uint32_t)w.pool.size(), 0 };
					const uint64_t* noff2 = j->store2->name_off.data();
					const uint64_t off1[2] = { noff[w.read], noff[w.read + 1] }, off2[2] = { noff2[w.read], noff2[w.read + 1] };
					bt_io_format_pairs(one1, names, off1, one2, j->store2->names.data(), off2, hw, refs, O.out, 0, 1, &text, &t);
					return;
				}
				if (sg.wide < 0) { bt_io_format(j->rb, names, noff, hb, refs, O.out, sg.lo, sg.hi, &text, &t); return; }
				Job::Wide& w = j->wide[(size_t)sg.wide];
				/* a one-read view whose slot count holds every hit of that read */
				bt_read_batch one = j->rb;
				one.n_reads = 1;
				one.seq = j->rb.seq + (size_t)w.read * j->rb.stride; one.qual = j->rb.qual + (size_t)w.read * j->rb.stride;
				one.len = j->rb.len + w.read; one.seed = j->rb.seed + w.read;
				bt_hit_batch hw = { w.hit_cap, w.hits.data(), &w.n_hits, &w.status, w.pool.data(), (uint32_t)w.pool.size(), 0 };
				const uint64_t off[2] = { noff[w.read], noff[w.read + 1] };
				bt_io_format(one, names, off, hw, refs, O.out, 0, 1, &text, &t);
			};
			auto run = [&](size_t si) {
				std::string text;
				text.swap(parts[si]);
				bt_out_tally t = {0, 0, 0, 0, 0, 0};
				run_into(si, text, t);
				parts[si].swap(text);
				tl[si] = t;
			};
			double t_wait = 0;                                   /* this thread waiting for a piece's text */
			if (TF > 1 && segs.size() > 1) {
				/* pieces are formatted in order of appearance by TF threads and written, in order, by this one as they finish */
				std::mutex m; std::condition_variable cv;
				std::vector<char> ready(segs.size(), 0);
				if (!pool) pool.reset(new WorkPool(TF));
				pool->start(segs.size(), [&](size_t si) {
					run(si);
					{ std::lock_guard<std::mutex> l(m); ready[si] = 1; }
					cv.notify_all();
				});
				for (size_t si = 0; si < segs.size(); si++) {
					const double tw = now_s();
					{ std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return ready[si] != 0; }); }
					t_wait += now_s() - tw;
					put_text(fout, parts[si].data(), parts[si].size());
				}
				pool->wait();
			} else for (size_t si = 0; si < segs.size(); si++) { run(si); put_text(fout, parts[si].data(), parts[si].size()); }
			const double tf = now_s();
			for (size_t si = 0; si < segs.size(); si++) {
				tally.aligned += tl[si].aligned; tally.unaligned += tl[si].unaligned; tally.maxed += tl[si].maxed; tally.reported += tl[si].reported;
				tally.sample_max |= tl[si].sample_max; tally.reported_paired += tl[si].reported_paired;
			}
			if (getenv("BT_IO_PROFILE")) fprintf(stderr, "[io] batch of %u: format on %d threads + write %.3f s (%zu pieces; %.3f s of it waiting for text)\n", n, TF, tf - tb, segs.size(), t_wait);
			if (dumping) {
				/* HitSink::dumpAlign / dumpUnal / dumpMaxed (hit.h:385-488): the read's record as it stood in
				 * the input; files are created when the first read goes to them; without --max, reads over
				 * the -m ceiling go to --un */
				const BtHostBatch& sb = *j->store;
				/* pairs: each mate's record to <name>_1<.ext> / <name>_2<.ext> (HitSink::openOf, hit.h:629-660) -- except
				 * for --12 input, whose one record per pair goes to <name> as it is (onePairFile_); the -m ceiling counts
				 * mate alignments, two per pair */
				const bool two_files = O.paired && !tabbed;
				const uint32_t ceiling = O.paired ? (O.pol.mhits == 0xffffffffu ? 0xffffffffu : O.pol.mhits * 2u) : O.pol.mhits;
				size_t wi = 0;
				for (uint32_t i = 0; i < n; i++) {
					uint32_t tot = j->n_hits[i];
					while (wi < j->wide.size() && j->wide[wi].read < i) wi++;
					if (wi < j->wide.size() && j->wide[wi].read == i) tot = j->wide[wi].n_hits;
					FILE** f; const std::string* nm;
					if (tot == 0) { f = &f_un; nm = &O.dump_un; }
					else if (tot > ceiling) { if (!O.dump_max.empty()) { f = &f_max; nm = &O.dump_max; } else { f = &f_un; nm = &O.dump_un; } }
					else { f = &f_al; nm = &O.dump_al; }
					if (nm->empty()) continue;
					FILE** f2 = f == &f_un ? &f_un2 : f == &f_max ? &f_max2 : &f_al2;
					if (!*f) {
						const size_t dot = nm->find_last_of('.');
						const std::string n1 = !two_files ? *nm : dot == std::string::npos ? *nm + "_1" : nm->substr(0, dot) + "_1" + nm->substr(dot);
						const std::string n2 = dot == std::string::npos ? *nm + "_2" : nm->substr(0, dot) + "_2" + nm->substr(dot);
						*f = fopen(n1.c_str(), "wb");
						if (two_files) *f2 = fopen(n2.c_str(), "wb");
						if (!*f || (two_files && !*f2)) { if (fatal.empty()) fatal = "Error: could not open read dump file " + *nm; abort_run.store(true); break; }
					}
					fwrite(sb.raw.data() + sb.raw_off[i], 1, (size_t)(sb.raw_off[i + 1] - sb.raw_off[i]), *f);
					if (two_files) { const BtHostBatch& s2 = *j->store2; fwrite(s2.raw.data() + s2.raw_off[i], 1, (size_t)(s2.raw_off[i + 1] - s2.raw_off[i]), *f2); }
				}
			}
			busy_write += now_s() - tb;
			g_tl.mark("write: done", j->seq);
			j->wide.clear();
			{
				const bool keep = !input_done.load();
				std::unique_ptr<ResultBufs> rbuf;
				if (will_stream && keep) {
					rbuf.reset(new ResultBufs());
					rbuf->hits.swap(j->hits); rbuf->n_hits.swap(j->n_hits); rbuf->status.swap(j->status); rbuf->mm_pool.swap(j->mm_pool);
					if (spare_res.try_put(rbuf)) rbuf.reset();
				}
				std::unique_ptr<BtHostBatch> st = std::move(j->store);
				if (keep && spare.try_put(st)) st.reset();
				let_go(j, rbuf, st);
			}
		}
		};
		write_batches();
		if (!full_teardown) (void)pool.release();            /* its threads go with the process: joining 128 of them is 0.1 s */
		else pool.reset();
		/* the text buffers (2.5 GB for batches of 12 M reads) are unmapped beside the closing of the files, not before it */
		if (!parts.empty()) { std::unique_ptr<Trash> t(new Trash()); t->text.swap(parts); to_reap.put(std::move(t)); }
	});

	/* ---- stage 2: the GPU.  Each searcher owns a context (its own stream and scratch); with two, one
	 * batch's long-running stragglers finish while the next batch already fills the machine ---- */
	const double t_search = now_s();
	std::vector<std::thread> searchers;
	for (int g = 0; g < G; g++) searchers.emplace_back([&, g] {
		if (!streamed) {
			for (;;) {
				std::unique_ptr<Job> j = to_gpu.take();
				if (j->last) { to_out.put(std::move(j)); return; }
				if (!abort_run.load()) {
					const double tb = now_s();
					if (j->rb.n_reads) j->error = search_job(ctxs[(size_t)g], O, j.get());
					if (j->unp && j->error.empty()) j->error = search_job(unp_ctxs[(size_t)g], OU, j->unp.get());
					busy_gpu[(size_t)g] += now_s() - tb;
				}
				to_out.put(std::move(j));
			}
		}
		/* Streamed: this thread keeps one context fed (bt_align_stream_*, carry-over on): batches go in one after the
		 * other; a batch comes out when the last of its reads is done -- the long-running ones ride along with the
		 * batches behind it -- and is post-processed on the thread's second context while the GPU works on. */
		bt_ctx* cs = ctxs[(size_t)g];
		bt_ctx* cr = redo_ctxs[(size_t)g];
		std::deque<std::unique_ptr<Job>> fl;                   /* in flight, oldest first */
		size_t my_fl = fl_lim;                                 /* this searcher's own: its device may have less room than asked for */
		int my_ticks = n_ticks;
		bool sized = false;
		auto drain = [&](int flush) {
			while (!fl.empty()) {
				void* tag = nullptr;
				const int rc = bt_align_stream_collect(cs, &tag, flush);
				if (rc == BT_OK && !tag) return;                /* the oldest is not complete yet */
				std::unique_ptr<Job> p = std::move(fl.front());
				fl.pop_front();
				g_tl.mark("search: results back", p->seq);
				if (rc != BT_OK || tag != (void*)p.get()) p->error = std::string("Error: search failed: ") + bt_strerror(rc != BT_OK ? rc : BT_ERR_DEVICE);
				else {
					int st = BT_OK;
					for (uint32_t i = 0; i < p->rb.n_reads && st == BT_OK; i++) if (p->status[i] & BT_ST_TOOSHORT) st = BT_ERR_READ_SHORT;
					p->error = search_finish(cr, O, p.get(), st, true);
				}
				const uint64_t sq = p->seq;
				to_out.put(std::move(p));
				g_tl.mark("search: handed to the writer's queue", sq);
			}
		};
		for (;;) {
			std::unique_ptr<Job> j = to_gpu.take();
			const double tb = now_s();
			if (j->last || abort_run.load()) {
				g_tl.mark("search: end of input, finishing what is parked", j->seq);
				/* the batches still in flight come out one by one: a tick lets what is parked run on for a while and says what
				 * it completed, so the writer works through the oldest batches while the stragglers of the later ones still run
				 * (a flush hands all of them over at once, when the last straggler of the last batch is done: 13 batches,
				 * 3.9 s of formatting and writing that overlapped nothing in round 4's timeline).  After as many ticks as a
				 * read may ride along, whatever is left is flushed. */
				if (!abort_run.load() && !getenv("BT_CLI_NO_TICKS")) {
					for (int tick = 0; tick < my_ticks && !fl.empty(); tick++) {
						drain(0);
						if (fl.empty()) break;
						if (bt_align_stream_tick(cs, 0) != BT_OK) break;
						g_tl.mark("search: tick", j->seq);
						/* until the oldest batch comes out, or for as long as a tick may take */
						const size_t before = fl.size();
						const double t0 = now_s();
						while (fl.size() == before && now_s() - t0 < 0.6) { drain(0); if (fl.size() == before) std::this_thread::sleep_for(std::chrono::microseconds(500)); }
					}
				}
				drain(1);
				busy_gpu[(size_t)g] += now_s() - tb;
				const bool end = j->last;
				to_out.put(std::move(j));
				if (end) return;
				continue;
			}
			/* With carry-over a batch is complete when the launches behind it have finished what it left running: up to twelve
			 * of them.  At the library's limit of batches in flight (BT_BATCH_RING - 2) the oldest is therefore waited for --
			 * the launches it depends on are all enqueued -- not forced out: forcing (flush) stops the wavefronts from taking
			 * new reads until the stragglers are done, which is what a 64 M-read run spent 2.8 s of its 18 on in round 4's
			 * timeline. */
			{
				const double tw = now_s();
				while (fl.size() >= my_fl && !abort_run.load()) {
					drain(0);
					if (fl.size() < my_fl) break;
					/* the oldest batch completes by itself within the time its launches take; if it has not after a minute
					 * something is wrong on the device: a flush waits for the stream and hands the error out */
					if (now_s() - tw > 60.0) { drain(1); break; }
					std::this_thread::sleep_for(std::chrono::microseconds(500));
				}
			}
			g_tl.mark("search: taken", j->seq);
			search_prepare(O, j.get());
			g_tl.mark("search: result arrays ready", j->seq);
			if (!sized) {
				/* the first batch says how large a batch in flight is on the device (-k and -a make its result slots several
				 * times the reads' size): no more of them ride than half the free HBM holds, and reads ride two launches fewer */
				sized = true;
				uint32_t room = 0;
				if (bt_align_stream_room(cs, &j->rb, &j->hb, &room) == BT_OK && (size_t)room < my_fl) {
					my_fl = room < 3u ? 3u : (size_t)room;
					const int cage = (int)my_fl - 2;
					if (bt_ctx_set_carry(cs, cage) != BT_OK) { j->error = "Error: bt_ctx_set_carry failed"; }
					my_ticks = cage + 2;
					if (O.timing) fprintf(stderr, "The device has room for %u batches of this size in flight: reads ride %d launches, %zu batches in flight\n", room, cage, my_fl);
				}
			}
			const int rc = bt_align_stream_submit(cs, &j->rb, &j->hb, j.get());
			g_tl.mark("search: submitted (uploaded, launch enqueued)", j->seq);
			if (rc != BT_OK) {
				drain(1);
				j->error = std::string("Error: search failed: ") + bt_strerror(rc);
				to_out.put(std::move(j));
				continue;
			}
			fl.push_back(std::move(j));
			drain(0);
			busy_gpu[(size_t)g] += now_s() - tb;
		}
	});
	for (auto& x : searchers) x.join();
	reader.join();
	writer.join();
	g_tl.mark("teardown: the stages' threads are gone", 0);
	for (int i = 0; i < n_reapers; i++) to_reap.put(std::unique_ptr<Trash>());
	if (O.timing) {
		print_timer("Time searching: ", now_s() - t_search);
		double bg = 0; for (double v : busy_gpu) bg += v;
		fprintf(stderr, "Stage busy time (s): read+parse %.2f, search (GPU, incl. PCIe; %d in flight) %.2f, format+write %.2f\n", busy_read, G, bg, busy_write);
	}
	fflush(fout);
	if (fout != stdout) fclose(fout);
	if (f_al) fclose(f_al);
	if (f_un) fclose(f_un);
	if (f_max) fclose(f_max);
	if (f_al2) fclose(f_al2);
	if (f_un2) fclose(f_un2);
	if (f_max2) fclose(f_max2);
	g_tl.mark("teardown: output files closed", 0);
	/* The output is complete.  What the process still holds -- the reader's window, the contexts and the index in HBM, the batches
	 * the reapers have not got to -- goes with the process: letting it go by hand took 2.4 s of a 19.9 s run (GPU call 11), most
	 * of it the runtime's lock passing between this thread and the reapers.  A run that failed, or one asked to
	 * (BT_CLI_TEARDOWN=1: leak checks), takes everything down in order. */
	if (full_teardown || !fatal.empty()) {
		if (rs) bt_io_close(rs);
		if (rs2) bt_io_close(rs2);
		g_tl.mark("teardown: inputs closed", 0);
		for (bt_ctx* c : ctxs) bt_ctx_destroy(c);
		for (bt_ctx* c : redo_ctxs) bt_ctx_destroy(c);
		for (bt_ctx* c : unp_ctxs) bt_ctx_destroy(c);
		g_tl.mark("teardown: contexts destroyed", 0);
		for (bt_index* x : idxs) bt_index_free(x);
		g_tl.mark("teardown: index freed", 0);
		for (auto& x : reapers) x.join();
		g_tl.mark("teardown: the last batches' memory let go", 0);
	}
	if (!fatal.empty()) { fprintf(stderr, "%s\n", fatal.c_str()); return 1; }
	if (!O.quiet) { std::string s; bt_io_summary(tally, &s); fputs(s.c_str(), stderr); }
	if (O.timing) print_timer("Overall time: ", now_s() - t_all);
	/* the reapers unmap side by side what the process's exit would unmap alone (they no longer hand page-locked arrays back) */
	if (!full_teardown) { for (auto& x : reapers) x.join(); g_tl.mark("teardown: the last batches' memory let go", 0); }
	g_tl.mark("end", 0);
	g_tl.print();
	/* every file is closed and every stream flushed: leave without the runtime's own teardown (the HIP runtime unloading its
	 * code objects and tearing down queues was most of a second at the end of every run) */
	remove_spooled();                                         /* (atexit handlers do not run) */
	fflush(nullptr);
	_exit(0);
}
