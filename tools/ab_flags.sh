#!/bin/bash
# A/B any micro-benchmark under extra -D flags: BENCH="tools/bench_qkv_time.py 224" tools/ab_flags.sh "<flags variant 1>" "<flags variant 2>" ...
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
    SYNCHFORMER_HIP_LIB=/tmp/libsf_ab$i.so python $R/$BENCH 2>&1 | grep -v amdgpu.ids
  done
done
