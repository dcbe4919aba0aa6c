#!/bin/bash
# kernel trace (per-dispatch rows with queue ids and timestamps) of the Stage-1 workload: tools/trace_stage1.sh <tag> -> gpurun_out/<tag>_stage1_trace.csv
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_s1
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_s1 -- python $R/bench.py --workload stage1 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads > /tmp/trace_s1.log 2>&1
f=$(find /tmp/trace_s1 -name '*kernel_trace.csv' | head -1)
python - "$f" $R/gpurun_out/${TAG}_stage1_trace.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
keep = rows[len(rows) * 3 // 5:]          # the last two of the five steps
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f)
    w.writerow(['queue', 'start_ns', 'end_ns', 'kernel'])
    t0 = int(keep[0]['Start_Timestamp'])
    for r in keep:
        w.writerow([r.get('Queue_Id', ''), int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name'][:80]])
print(len(rows), 'dispatches;', len(keep), 'kept')
P
grep -h '"metric"' /tmp/trace_s1.log | tail -1 | cut -c1-200
