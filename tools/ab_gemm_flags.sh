#!/bin/bash
# A/B the GEMM shapes under extra -D flags: tools/ab_gemm_flags.sh "<flags variant 1>" "<flags variant 2>" ...   (SEGS="112 224", BENCH=tools/bench_gemm_mx.py for the MXFP8 kernel)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags $R/synchformer_amd/csrc/*.hip -o /tmp/libsf_ab$i.so || exit 1
done
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    echo "=== variant $i: '$flags' (rep $rep)"
    SYNCHFORMER_HIP_LIB=/tmp/libsf_ab$i.so CFGS=${CFGS:-7} python $R/${BENCH:-tools/bench_gemm.py} ${SEGS:-112} 2>&1 | grep -v amdgpu.ids
  done
done
