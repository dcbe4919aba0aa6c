#!/bin/bash
# A/B the fused GEMM + residual + LayerNorm kernel under extra -D flags: tools/ab_gemm_ln_flags.sh "<flags variant 1>" "<flags variant 2>" ...
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
    SYNCHFORMER_HIP_LIB=/tmp/libsf_ab$i.so python $R/tools/bench_gemm_ln.py ${SEGS:-224} 2>&1 | grep fused | sed "s/.*| fused/fused/"
  done
done
