#!/bin/bash
# rocprofv3 kernel traces of the three train-side workloads of bench.py (run on the GPU box through gpurun):
#   tools/profile_workloads.sh <tag>   -> gpurun_out/<tag>_{stage1,train,ft}_kernel_stats.csv + the bench lines in gpurun_out/<tag>_workloads.log
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_workloads.log
for wl in "stage1 --batch 2" "train" "ft"; do
  set -- $wl; name=$1
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads > /tmp/prof_$name.log 2>&1
  grep -h '"metric"' /tmp/prof_$name.log | tail -1 >> $R/gpurun_out/${TAG}_workloads.log
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${TAG}_${name}_kernel_stats.csv
done
cat $R/gpurun_out/${TAG}_workloads.log | cut -c1-300
