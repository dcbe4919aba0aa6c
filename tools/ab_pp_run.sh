#!/bin/bash
# GPU box: interleaved runs of every variant built by tools/ab_pp.sh.  SEGS="224" CFGS=11 REPS=2 BENCH=tools/bench_gemm.py
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for rep in $(seq 1 ${REPS:-2}); do
  for so in $(ls $R/tools/ab_build/libsf_*.so | sort -V); do
    i=$(basename $so .so | sed s/libsf_//)
    echo "=== variant $i: '$(cat $R/tools/ab_build/flags_$i.txt)' (rep $rep)"
    SYNCHFORMER_HIP_LIB=$so CFGS=${CFGS:-11} KMAJOR=${KMAJOR:-0} python $R/${BENCH:-tools/bench_gemm.py} ${SEGS:-224} 2>&1 | grep -v amdgpu.ids
  done
done
