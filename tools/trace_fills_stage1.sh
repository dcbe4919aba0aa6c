#!/bin/bash
# Which fill launches does a Stage-1 step contain?  Per-dispatch trace -> (grid size, kernel before, kernel after) histogram of the FillFunctor launches of the last step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_f
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_f -- python $R/bench.py --workload stage1 --batch 2 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-workloads > /tmp/trace_f.log 2>&1
f=$(find /tmp/trace_f -name '*kernel_trace.csv' | head -1)
python - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_clip_kernel')]
rows = rows[ends[-2] + 1:ends[-1] + 1]
c = collections.Counter()
for i, r in enumerate(rows):
    if 'FillFunctor' in r['Kernel_Name']:
        ty = 'fill<8B>' if 'kernel<8' in r['Kernel_Name'] else 'fill<4B>'
        prev = rows[i - 1]['Kernel_Name'].split('(')[0][-40:] if i else '-'
        nxt = rows[i + 1]['Kernel_Name'].split('(')[0][-40:] if i + 1 < len(rows) else '-'
        c[(ty, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Queue_Id', ''), prev, nxt)] += 1
for k, v in c.most_common(40):
    print(v, k)
print(sum(c.values()), 'fills in the step;', len(rows), 'dispatches')
P
