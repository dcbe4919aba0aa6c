#!/bin/bash
# rocprofv3 kernel stats of the fine-tune workload with the two towers on one stream (kernel durations not stretched by co-running kernels)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ft
SF_AUDIO_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft -- python $R/bench.py --workload ft --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads > /tmp/prof_ft.log 2>&1
grep -h '"metric"' /tmp/prof_ft.log | tail -1 | cut -c1-220
cp "$(find /tmp/prof_ft -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${TAG}_ft_serial_kernel_stats.csv
