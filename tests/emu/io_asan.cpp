/*
 * io_asan.cpp -- TEST INFRASTRUCTURE ONLY.  The read parsers of bowtie_amd/csrc/bt_io.cpp built with AddressSanitizer and
 * UBSan and run over files -- well-formed or not -- in small batches: memory errors in the C++ parsers show up here.
 *
 *   usage: io_asan <format 0..5> <flags> <trim5> <trim3> <batch reads> <file> [<file> ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../bowtie_amd/csrc/bt_io.h"

int main(int argc, char** argv)
{
	if (argc < 7) return 2;
	bt_read_opts o;
	memset(&o, 0, sizeof(o));
	o.format = atoi(argv[1]); o.flags = (uint32_t)atoi(argv[2]); o.trim5 = atoi(argv[3]); o.trim3 = atoi(argv[4]);
	if (o.format == BT_FMT_FASTA_CONT) { o.cont_len = 7; o.cont_freq = 3; }
	const uint32_t batch = (uint32_t)atoi(argv[5]);
	unsigned long long reads = 0, bases = 0;
	for (int i = 6; i < argc; i++) {
		std::string err;
		BtReadStream* s = bt_io_open(argv[i], o, &err);
		BtHostBatch b;
		for (;;) {
			const int rc = bt_io_next(s, batch, 2, &b, &err);
			if (rc != BT_OK || b.n == 0) break;
			reads += b.n;
			for (uint32_t k = 0; k < b.n; k++) bases += b.len[k];
		}
		bt_io_close(s);
	}
	printf("%llu reads, %llu bases\n", reads, bases);
	return 0;
}
