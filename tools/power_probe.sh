#!/bin/bash
# Runs tools/probe/mfma_power.bin in its four modes and samples rocm-smi (power, sclk) next to each: the MFMA rate the package power cap leaves.
#   tools/power_probe.sh <tag>  -> gpurun_out/power_probe_<tag>.txt
tag=${1:-r03}; out=gpurun_out/power_probe_${tag}.txt; mkdir -p gpurun_out; : > $out
IFS=";" read -ra modes <<< "${MODES:-bf16 rand 1 4;bf16 rand 2 4;bf16 zero 2 4;mx rand 1 4;mx rand 2 4;mx zero 2 4}"
for mode in "${modes[@]}"; do
    echo "== $mode" >> $out
    tools/probe/mfma_power.bin $mode > gpurun_out/.mfma_power.log 2>&1 &
    pid=$!
    while kill -0 $pid 2>/dev/null; do
        rocm-smi -P -c 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | awk '{printf "%s ", $NF} END {print ""}' >> $out
        sleep 0.4
    done
    wait $pid
    tail -3 gpurun_out/.mfma_power.log >> $out
done
cat $out
