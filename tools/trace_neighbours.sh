#!/bin/bash
# Which launches of a bench.py workload match <pattern>, and between which kernels do they sit?  tools/trace_neighbours.sh <workload> <pattern> [bench args...]
WL=${1:-infer}; PAT=${2:-copyBuffer}; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_n
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_n -- python $R/bench.py --workload $WL --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads "$@" > /tmp/trace_n.log 2>&1
f=$(find /tmp/trace_n -name '*kernel_trace.csv' | head -1)
python - "$f" "$PAT" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) * 3 // 4:]                                  # the last of the four steps
c = collections.Counter()
short = lambda r: r['Kernel_Name'].split('(')[0][-44:]
for i, r in enumerate(rows):
    if sys.argv[2] in r['Kernel_Name']:
        c[(r.get('Grid_Size_X', '?'), r.get('Queue_Id', ''), short(rows[i - 1]) if i else '-', short(rows[i + 1]) if i + 1 < len(rows) else '-')] += 1
for k, v in c.most_common(30):
    print(v, k)
print(sum(c.values()), 'matching launches of', len(rows), 'dispatches in the last step')
P
