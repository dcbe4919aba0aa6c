#!/bin/bash
# Build A/B variants of the quadrant-phased GEMM (config 11) under extra -D flags, in the build container:
#     tools/ab_pp.sh "<flags variant 1>" "<flags variant 2>" ...      ("" = product build)
# -> tools/ab_build/libsf_<i>.so (+ flags_<i>.txt); run them on the GPU box with tools/ab_pp_run.sh
R=$(cd $(dirname $0)/.. && pwd)
python -m synchformer_amd.build >/dev/null || exit 1
mkdir -p $R/tools/ab_build && rm -f $R/tools/ab_build/*
SRC=${SRC:-sf_gemm_pp}
i=0
for flags in "$@"; do
  i=$((i+1))
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $R/synchformer_amd/csrc/$SRC.hip -o /tmp/ab_$i.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/synchformer_amd/lib/obj/*.o | grep -v /$SRC.o) /tmp/ab_$i.o -o $R/tools/ab_build/libsf_$i.so &&
    echo "$flags" > $R/tools/ab_build/flags_$i.txt ) &
done
wait
ls -la $R/tools/ab_build
