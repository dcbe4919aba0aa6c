#!/bin/bash
# rocprofv3 kernel trace of the Stage-1 (AVCLIP) train step; summary -> gpurun_out/stage1_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s1 -- python $R/bench.py --workload stage1 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/stage1_prof.log 2>&1
f=$(find /tmp/prof_s1 -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/stage1_kernel_stats.csv
