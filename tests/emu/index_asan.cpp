/*
 * index_asan.cpp -- TEST INFRASTRUCTURE ONLY.  The index loader of bowtie_amd/csrc/bt_host.cpp built with AddressSanitizer
 * and UBSan: loads <base> (fw, mirror, 2-bit reference) the way bt_index_load does, exceptions mapped to an error code as
 * there, and prints the result.  Fed damaged index files it must answer with an error code or a loaded index -- not crash.
 */
#include <stdio.h>
#include <exception>
#include "../../bowtie_amd/csrc/bt_host.h"

int main(int argc, char** argv)
{
	if (argc < 2) return 2;
	const std::string base = argv[1];
	int rc = BT_OK, rcm = BT_OK, rcr = BT_OK;
	BtIndexHost fw, mir;
	BtRefHost ref;
	try { rc = bt_host_index_load(base, true, -1, &fw); } catch (const std::exception&) { rc = BT_ERR_FORMAT; }
	if (rc == BT_OK) {
		try { rcm = bt_host_index_load(base + ".rev", false, -1, &mir); } catch (const std::exception&) { rcm = BT_ERR_FORMAT; }
		try { rcr = bt_host_ref_load(base, fw, &ref); } catch (const std::exception&) { rcr = BT_ERR_FORMAT; }
	}
	printf("fw %d mirror %d ref %d len %llu\n", rc, rcm, rcr, rc == BT_OK ? (unsigned long long)fw.len : 0ull);
	return 0;
}
