#!/bin/bash
# GPU box: tools/bench_qkv_time2.py over every variant built by `SRC=sf_qkv_time2 tools/ab_pp.sh ...` (interleaved repetitions); STAGGERS="0 1 2": SF_QT2_STAGGER values per variant
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for rep in $(seq 1 ${REPS:-2}); do
for so in $(ls $R/tools/ab_build/libsf_*.so | sort -V); do i=$(basename $so .so | sed s/libsf_//);
for st in ${STAGGERS:-0}; do echo "== variant $i: $(cat $R/tools/ab_build/flags_$i.txt) stagger $st"; SF_QT2_STAGGER=$st SYNCHFORMER_HIP_LIB=$so python $R/tools/bench_qkv_time2.py ${SEGS:-224} 2>&1 | tail -1 | sed 's/.*kernels alone/kernels alone/'; done; done
done
