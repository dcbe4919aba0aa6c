#!/bin/bash
# the delivery ceiling measured by something other than the kernels under judgement (run on the GPU box).
#   tools/probe/l2_stream.bin csv       -> gpurun_out/r06/l2_stream.csv   (standalone L2 / Infinity-Cache / HBM -> LDS | VGPR streams)
#   tools/bench_library_gemm.py         -> gpurun_out/r06/library_gemm.txt (vendor library beside sf_gemm_bf16 config 11 on the four shapes)
#   tools/bench_qkv_space.py / bench_qkv_time2.py / bench_gemm_ln.py        (the fused launches on the same box)
set -u
out=gpurun_out/r06; mkdir -p $out
[ -x tools/probe/l2_stream.bin ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/l2_stream.hip -o tools/probe/l2_stream.bin
timeout 900 tools/probe/l2_stream.bin csv > $out/l2_stream.csv 2> $out/l2_stream.err
timeout 300 python tools/bench_library_gemm.py 224 > $out/library_gemm.txt 2>&1
TORCH_BLAS_PREFER_HIPBLASLT=1 timeout 300 python tools/bench_library_gemm.py 224 > $out/library_gemm_hipblaslt.txt 2>&1
timeout 300 python tools/bench_qkv_space.py 224 > $out/qkv_space.txt 2>&1
timeout 300 python tools/bench_qkv_time2.py 224 > $out/qkv_time2.txt 2>&1
timeout 300 python tools/bench_gemm_ln.py 224 > $out/gemm_ln.txt 2>&1
tail -3 $out/*.txt; wc -l $out/l2_stream.csv
