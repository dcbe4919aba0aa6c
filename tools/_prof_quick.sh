cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export SF_AUDIO_SIDE_STREAM=0
rm -rf /tmp/pq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o q -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/pq.log 2>&1
f=$(find /tmp/pq -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/quick_kernel_stats.csv
grep -h '"metric"' /tmp/pq.log | tail -1 | cut -c1-200
