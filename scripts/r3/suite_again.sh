#!/bin/bash
# the whole GPU suite and smoke() once more on the round's last device code
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
t0=$(date +%s)
timeout 420 python -m pytest tests -m gpu -q -x > $O/suite.txt 2>&1; echo "GPU suite at HEAD: $(tail -1 $O/suite.txt) [$(( $(date +%s) - t0 )) s wall]" | tee $O/SUMMARY.txt
grep -E "^FAILED|^ERROR" $O/suite.txt | head -5 | tee -a $O/SUMMARY.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/SUMMARY.txt
