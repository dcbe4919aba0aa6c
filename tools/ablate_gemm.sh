#!/bin/bash
# Build throwaway ablation variants of the library (never shipped) and time the GEMM shapes with each.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for mask in ${MASKS:-1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSF_ABL=$mask $R/synchformer_amd/csrc/*.hip -o /tmp/libsf_abl$mask.so || exit 1
  echo "=== SF_ABL=$mask (1: no epilogue HBM traffic, 2: no operand loads after the prologue)"
  SYNCHFORMER_HIP_LIB=/tmp/libsf_abl$mask.so python $R/tools/bench_gemm.py ${1:-27} 2>&1 | grep -v amdgpu.ids
done
