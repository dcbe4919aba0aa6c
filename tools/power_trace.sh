#!/bin/bash
# Samples socket power, sclk and temperature with rocm-smi while bench.py's
# forward loop runs: evidence for (or against) the power-limited clock that
# DESIGN.md's GEMM-ceiling paragraph assumes.  Usage (on the GPU box):
#   tools/power_trace.sh <tag> [bench args...]    -> gpurun_out/power_<tag>.{log,txt}
tag=${1:-r03}; shift
out=gpurun_out; mkdir -p $out
rocm-smi --showmaxpower --showclocks -P > $out/power_${tag}_idle.txt 2>&1
python bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-workloads "$@" \
    > $out/power_${tag}_bench.log 2>&1 &
pid=$!
: > $out/power_${tag}.log
while kill -0 $pid 2>/dev/null; do
    echo "t=$(date +%s.%N)" >> $out/power_${tag}.log
    rocm-smi -P -c -t --showperflevel 2>/dev/null | grep -E "Power|sclk|fclk|mclk|Temperature \(Sensor (junction|edge)|Performance" >> $out/power_${tag}.log
    sleep 0.3
done
wait $pid
tail -1 $out/power_${tag}_bench.log
