#!/bin/bash
# Fourth GPU call of round 4: where does bt_best_kernel's wavefront time go?  The section timers of bt_best.h (-DBF_PROFILE,
# make bestprof) on the four best-first workloads, shipped engine and fast-extend-without-refill.
#   gpurun --timeout 600 -- 'bash scripts/r4/fourth.sh'
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
S=$O/SUMMARY.txt; : > $S
say() { echo "$*" | tee -a $S; }
run() {   # lib workload extra...
	local lib=$1 wl=$2; shift 2
	local f=$O/prof_${wl}_$lib
	BT_LIB=$lib timeout 280 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu --no-verify --also none "$@" > $f.json 2> $f.log
	say "== $lib $wl $*"
	grep "section" $f.log | sed 's/^\[bench\] //' | tee -a $S
	python -c "import json; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); print('   %.3f M reads/s, %.1f ms/step' % (d.get('reads_processed_per_s', d['value'])/1e6, d['ms_per_step']))" 2>&1 | tail -1 | tee -a $S
}
for lib in libbowtie_amd_bestprof_nr.so libbowtie_amd_bestprof.so; do
	run $lib ecoli_n2_best_100
	run $lib ecoli_pe_n1_best_50
done
for lib in libbowtie_amd_bestprof_nr.so libbowtie_amd_bestprof.so; do
	run $lib big_n2_best_100 --reads 4000000
	run $lib big_pe_n1_best_50 --reads 3000000
done
cat $S
