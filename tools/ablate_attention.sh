#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for mask in ${MASKS:-1 2 4 7}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSF_ATT_ABL=$mask $R/synchformer_amd/csrc/*.hip -o /tmp/libsf_att$mask.so || exit 1
  echo "=== SF_ATT_ABL=$mask (1 no stores, 2 no P V, 4 no exp)"
  SYNCHFORMER_HIP_LIB=/tmp/libsf_att$mask.so python $R/tools/bench_attention.py ${1:-27} 2>&1 | grep -v amdgpu.ids | head -1
done
