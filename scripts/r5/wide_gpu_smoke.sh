#!/bin/bash
# bowtie-amd-l (64-bit rows) on the GPU, no python: the reference-made family goldens (bowtie-align-l on multi_l), rows numbered
# across 2^32 over 1024-row segments (tests/test_wide_rows_emu.py).  Prints one line per case; all must say 'same'.
cd "$(dirname "$0")/../.."
export BT_TEST_KNOBS=1 BT_WIDE_ROW_BIAS=4294937600 BT_WIDE_SEG_SHIFT=4
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -n 2 -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn36.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = f9f627e8a58a52504fbc86d5a895ffda ] && echo "syn36__n2: same as bowtie-align-l" || echo "syn36__n2: DIFFERENT ($got)"
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -v 2 -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn100.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = 483f51acb970ec1d2057b0395922af39 ] && echo "syn100__v2: same as bowtie-align-l" || echo "syn100__v2: DIFFERENT ($got)"
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -n 3 -y -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn50lowq.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = 6f33de9b2d56bff9d0764585e56eaaf3 ] && echo "syn50lowq__n3_y: same as bowtie-align-l" || echo "syn50lowq__n3_y: DIFFERENT ($got)"
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -n 2 -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn150.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = 365a3ecd42c3165b6a9213d906252545 ] && echo "syn150__n2: same as bowtie-align-l" || echo "syn150__n2: DIFFERENT ($got)"
unset BT_WIDE_ROW_BIAS BT_WIDE_SEG_SHIFT
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -n 2 -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn36.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = f9f627e8a58a52504fbc86d5a895ffda ] && echo "syn36__n2 (rows as in the files): same as bowtie-align-l" || echo "syn36__n2 (rows as in the files): DIFFERENT ($got)"
got=$(timeout 60 bowtie_amd/bowtie-amd-l -S --sam-nohead -v 2 -x tests/golden/family/multi_l scripts/r5/wide_smoke/syn100.fq 2>/dev/null | md5sum | cut -d' ' -f1); [ "$got" = 483f51acb970ec1d2057b0395922af39 ] && echo "syn100__v2 (rows as in the files): same as bowtie-align-l" || echo "syn100__v2 (rows as in the files): DIFFERENT ($got)"
