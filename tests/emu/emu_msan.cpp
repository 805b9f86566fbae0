/*
 * emu_msan.cpp -- TEST INFRASTRUCTURE ONLY.  The host build of the automaton (bt_emu.cpp) under clang's
 * MemorySanitizer, with everything the HIP kernel leaves undefined poisoned: finds reads of uninitialised LDS,
 * scratch arena or result words that a zero-filled emulator run cannot see.
 *
 *   usage: emu_msan <index base> <v|n> <mms> <all 0/1> <khits> <lanes> <rl_mode> <read> [<read> ...]
 *   a <read> of the form @<file> names a file of "<bases> <qualities>" lines (Phred+33); EMU_FR_CAP / EMU_ENT_CAP /
 *   EMU_PAL_CAP set the arena capacities (default 64 / 768 / 1024: nothing overflows), EMU_MHITS the -m ceiling,
 *   EMU_SEEDLEN / EMU_QUAL_THRESH -l / -e; EMU_PRINT_HITS=1 prints every stored hit with its mismatch list
 */
#define EMU_MSAN 1
#include "bt_emu.cpp"
#include <stdlib.h>

int main(int argc, char** argv)
{
	if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
	void* ix = emu_index_load(argv[1], 1, -1);
	if (!ix) { fprintf(stderr, "cannot load %s\n", argv[1]); return 2; }
	bt_policy pol;
	memset(&pol, 0, sizeof(pol));
	pol.mode = BT_MODE_N; pol.mms = 2; pol.seed_len = 28; pol.qual_thresh = 70; pol.max_bts = 125;
	pol.maq_round = 1; pol.khits = 1; pol.mhits = 0xffffffffu; pol.max_ins = 250; pol.mate1_fw = 1; pol.pair_tries = 100;
	pol.mode = argv[2][0] == 'v' ? BT_MODE_V : BT_MODE_N;
	pol.mms = atoi(argv[3]); pol.all_hits = atoi(argv[4]); pol.khits = (uint32_t)atoi(argv[5]);
	const uint32_t lanes = (uint32_t)atoi(argv[6]), rl_mode = (uint32_t)atoi(argv[7]);
	/* (plain C strings: libstdc++'s own is not instrumented, and MemorySanitizer reports what it cannot see into) */
	std::vector<char*> rs, qs;
	auto add = [&](const char* r, const char* q) {
		const size_t l = strlen(r);
		char* rc = (char*)malloc(l + 1); memcpy(rc, r, l + 1);
		char* qc = (char*)malloc(l + 1);
		for (size_t k = 0; k < l; k++) qc[k] = (q && k < strlen(q)) ? q[k] : 'I';
		qc[l] = 0;
		rs.push_back(rc); qs.push_back(qc);
	};
	for (int a = 8; a < argc; a++) {
		if (argv[a][0] == '@') {
			FILE* f = fopen(argv[a] + 1, "r");
			if (!f) { fprintf(stderr, "cannot open %s\n", argv[a] + 1); return 2; }
			static char line[8192], r[4096], q[4096];
			while (fgets(line, sizeof(line), f)) {
				r[0] = q[0] = 0;
				const int k = sscanf(line, "%4095s %4095s", r, q);
				if (k >= 1) add(r, k >= 2 ? q : nullptr);
			}
			fclose(f);
		} else add(argv[a], nullptr);
	}
	const uint32_t n = (uint32_t)rs.size();
	uint32_t maxLen = 0;
	for (uint32_t i = 0; i < n; i++) { const uint32_t l = (uint32_t)strlen(rs[i]); if (l > maxLen) maxLen = l; }
	const uint32_t stride = (maxLen + 15u) & ~15u;
	/* the rows' padding is undefined on the device too (the caller only writes len bytes) */
	uint8_t* seq = (uint8_t*)malloc((size_t)n * stride + 64); uint8_t* qual = (uint8_t*)malloc((size_t)n * stride + 64);
	std::vector<uint16_t> len(n); std::vector<uint32_t> seed(n);
	for (uint32_t i = 0; i < n; i++) {
		const char* r = rs[i];
		len[i] = (uint16_t)strlen(rs[i]); seed[i] = 12345u + i;
		for (uint32_t k = 0; k < len[i]; k++) {
			const char c = r[k];
			seq[(size_t)i * stride + k] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
			qual[(size_t)i * stride + k] = (uint8_t)qs[i][k];
		}
	}
	auto envu = [](const char* k, uint32_t d) { const char* v = getenv(k); return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : d; };
	pol.mhits = envu("EMU_MHITS", pol.mhits); pol.seed_len = envu("EMU_SEEDLEN", pol.seed_len); pol.qual_thresh = envu("EMU_QUAL_THRESH", pol.qual_thresh);
	const uint32_t hitCap = 64;
	bt_hit* hits = (bt_hit*)malloc(sizeof(bt_hit) * n * hitCap);
	uint32_t* nh = (uint32_t*)calloc(n, 4); uint8_t* st = (uint8_t*)calloc(n, 1);
	uint16_t* mm = (uint16_t*)malloc(2u * n * hitCap * 8);
	bt_read_batch in; memset(&in, 0, sizeof(in));
	in.n_reads = n; in.stride = stride; in.seq = seq; in.qual = qual; in.len = len.data(); in.seed = seed.data();
	bt_hit_batch out; memset(&out, 0, sizeof(out));
	out.hit_cap = hitCap; out.hits = hits; out.n_hits = nh; out.status = st; out.mm_pool = mm; out.mm_pool_cap = n * hitCap * 8;
	const int rc = emu_align_batch(ix, &pol, &in, &out, nullptr, lanes, envu("EMU_FR_CAP", 64), envu("EMU_ENT_CAP", 12 * 64), envu("EMU_PAL_CAP", 1024), rl_mode);
	printf("rc %d\n", rc);
	const bool ph = envu("EMU_PRINT_HITS", 0) != 0;
	for (uint32_t i = 0; i < n; i++) {
		printf("read %u: %u hits status %u\n", i, nh[i], st[i]);
		if (!ph || (st[i] & 8u)) continue;
		/* (the values are printed, i.e. branched on: an undefined hit field or mismatch entry is reported here) */
		for (uint32_t k = 0; k < nh[i] && k < hitCap; k++) {
			const bt_hit& h = hits[(size_t)i * hitCap + k];
			printf("  hit %u:%u %c cost %u stratum %u oms %u mms", h.tidx, h.toff, h.fw ? '+' : '-', h.cost, h.stratum, h.oms);
			for (uint32_t j = 0; j < h.nmm; j++) printf(" %u:%u", mm[h.mm_off + j] & 0x3ffu, (mm[h.mm_off + j] >> 12) & 3u);
			printf("\n");
		}
	}
	return 0;
}
