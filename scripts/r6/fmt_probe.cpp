// Diagnostic (round 6, GPU call 13): does bt_io_format scale with threads on the GPU box's host the way bowtie-amd's writer uses it?
// A batch of N synthetic 100 bp reads with one alignment each, formatted to SAM in 4 x T pieces by T threads (pieces handed out by a
// counter, buffers kept from repetition to repetition), for T = 1 .. 256.  Built against bowtie_amd/csrc/bt_io.o:
//   g++ -O3 -std=c++17 -Ibowtie_amd/csrc scripts/r6/fmt_probe.cpp bowtie_amd/csrc/bt_io.o -lz -lpthread -o /tmp/fmt_probe
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include "bt_io.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* big(size_t bytes, bool huge) {
	const size_t H = 2u << 20, len = ((bytes + H - 1) & ~(H - 1)) + H;
	char* m = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	char* a = (char*)(((uintptr_t)m + H - 1) & ~(uintptr_t)(H - 1));
	if (huge) madvise(a, len - (a - m), MADV_HUGEPAGE);
	return a;
}
int main(int argc, char** argv) {
	const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 12582912u, L = 100, stride = 112;
	const bool huge = argc > 2 ? atoi(argv[2]) != 0 : true;
	uint8_t* seq = (uint8_t*)big((size_t)n * stride, huge); uint8_t* qual = (uint8_t*)big((size_t)n * stride, huge);
	uint16_t* len = (uint16_t*)big((size_t)n * 2, huge); uint32_t* seed = (uint32_t*)big((size_t)n * 4, huge);
	bt_hit* hits = (bt_hit*)big((size_t)n * sizeof(bt_hit), huge); uint32_t* nh = (uint32_t*)big((size_t)n * 4, huge);
	uint8_t* st = (uint8_t*)big(n, huge); uint16_t* pool = (uint16_t*)big((size_t)n * 2, huge);
	std::vector<uint64_t> off((size_t)n + 1);
	char* names = (char*)big((size_t)n * 16, huge);
	{
		// filled by 64 threads, as the parser's threads fill a batch
		for (uint32_t i = 0; i <= n; i++) off[i] = (uint64_t)i * 12;
		std::vector<std::thread> th;
		for (int t = 0; t < 64; t++) th.emplace_back([&, t] {
			uint32_t x = 12345u + (uint32_t)t;
			for (uint32_t i = (uint32_t)((uint64_t)n * t / 64); i < (uint32_t)((uint64_t)n * (t + 1) / 64); i++) {
				for (uint32_t k = 0; k < stride; k++) { x = x * 1664525u + 1013904223u; seq[(size_t)i * stride + k] = k < L ? (x >> 20) & 3 : 4; qual[(size_t)i * stride + k] = k < L ? 33 + ((x >> 8) % 40) : '!'; }
				len[i] = L; seed[i] = x; snprintf(names + (size_t)i * 12, 13, "r%011u", i);
				memset(&hits[i], 0, sizeof(bt_hit)); hits[i].tidx = i % 24; hits[i].toff = x >> 4; hits[i].fw = i & 1; hits[i].nmm = (i % 3 == 0); hits[i].mm_off = i;
				pool[i] = (uint16_t)((i % 90) | (1 << 10)); nh[i] = (i % 5 == 0) ? 0 : 1; st[i] = 0;
			}
		});
		for (auto& x : th) x.join();
	}
	bt_read_batch rb; rb.n_reads = n; rb.stride = stride; rb.seq = seq; rb.qual = qual; rb.len = len; rb.seed = seed;
	bt_hit_batch hb = { 1, hits, nh, st, pool, n, n };
	BtRefNames refs; for (int i = 0; i < 24; i++) { refs.names.push_back("chr" + std::to_string(i)); refs.lens.push_back(100000000); }
	bt_out_opts o; memset(&o, 0, sizeof(o)); o.sam = 1; o.khits = 1; o.mhits = 0xffffffffu;
	for (int T : {1, 16, 32, 64, 128, 256}) {
		const size_t P = T == 1 ? 8 : (size_t)T * 4;
		std::vector<std::string> parts(P);
		for (auto& s : parts) s.reserve((size_t)(n / P + 1) * 340);
		double best = 1e9, first = 0;
		for (int rep = 0; rep < 4; rep++) {
			for (auto& s : parts) s.clear();
			std::atomic<size_t> next(0);
			const double t0 = now();
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++) th.emplace_back([&] {
				for (;;) {
					const size_t si = next++;
					if (si >= P) return;
					std::string text; text.swap(parts[si]);
					bt_out_tally tl = {0, 0, 0, 0, 0, 0};
					bt_io_format(rb, names, off.data(), hb, refs, o, (uint32_t)((uint64_t)n * si / P), (uint32_t)((uint64_t)n * (si + 1) / P), &text, &tl);
					parts[si].swap(text);
				}
			});
			for (auto& x : th) x.join();
			const double s = now() - t0;
			if (rep == 0) first = s;
			if (s < best) best = s;
		}
		size_t bytes = 0; for (auto& s : parts) bytes += s.size();
		printf("%3d threads, %zu pieces: first pass %.3f s, best of the next three %.3f s = %.1f M reads/s, %.1f GB/s of text (%s pages)\n", T, P, first, best, n / best / 1e6, bytes / best / 1e9, huge ? "huge" : "4 KB");
		fflush(stdout);
	}
	return 0;
}
