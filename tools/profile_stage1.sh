#!/bin/bash
# rocprofv3 kernel stats of the Stage-1 workload only: tools/profile_stage1.sh <tag> -> gpurun_out/<tag>_stage1_kernel_stats.csv
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s1 -- python $R/bench.py --workload stage1 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads > /tmp/prof_s1.log 2>&1
grep -h '"metric"' /tmp/prof_s1.log | tail -1 | cut -c1-220
cp "$(find /tmp/prof_s1 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${TAG}_stage1_kernel_stats.csv
