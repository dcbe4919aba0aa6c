#!/bin/bash
# Build a throw-away copy of the WHOLE library under extra -D flags (a flag that reaches every source through sf_common.h), in the build container:
#     tools/ab_full.sh <name> "<flags>"   -> tools/ab_build/libsf_<name>.so     (run against it with SYNCHFORMER_HIP_LIB=tools/ab_build/libsf_<name>.so)
R=$(cd $(dirname $0)/.. && pwd)
name=$1; flags=$2
obj=/tmp/ab_full_$name; mkdir -p $obj $R/tools/ab_build
for src in $R/synchformer_amd/csrc/*.hip; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $src -o $obj/$(basename $src .hip).o ) &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o $R/tools/ab_build/libsf_$name.so && echo "$flags" > $R/tools/ab_build/flags_$name.txt && ls -la $R/tools/ab_build/libsf_$name.so
