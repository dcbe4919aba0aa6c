#!/bin/bash
# Build the library with extra -D flags into /tmp and run the GEMM unit tests against it: tools/ab_test_flags.sh "<flags>"
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $1 $R/synchformer_amd/csrc/*.hip -o /tmp/libsf_abt.so || exit 1
SYNCHFORMER_HIP_LIB=/tmp/libsf_abt.so python -m pytest $R/tests/test_kernels_gpu.py -x -q -k "gemm and (cfg7 or auto)" 2>&1 | tail -2
