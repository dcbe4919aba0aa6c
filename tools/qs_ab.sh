#!/bin/bash
# GPU box: tools/bench_qkv_space.py over every variant built by `SRC=sf_qkv_space tools/ab_pp.sh ...` (interleaved repetitions)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for rep in $(seq 1 ${REPS:-2}); do
for so in $(ls $R/tools/ab_build/libsf_*.so | sort -V); do i=$(basename $so .so | sed s/libsf_//); echo "== variant $i: $(cat $R/tools/ab_build/flags_$i.txt)"; SYNCHFORMER_HIP_LIB=$so python $R/tools/bench_qkv_space.py ${SEGS:-224} 2>&1 | tail -1 | sed 's/.*fused (side/fused (side/'; done
done
