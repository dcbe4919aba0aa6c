#!/bin/bash
# per-kernel view of the one-clip forward (BASELINE configs[0]): rocprofv3 --kernel-trace --stats, both towers on one stream (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
TAG=${1:-b1}
OUT=$R/gpurun_out/r06/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SF_AUDIO_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b1 -- python $R/tools/b1_forward.py 20 ${2:-1} > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/b1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time per forward: {tot/23/1e6:.3f} ms over 23 forwards')
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/23/1e3:9.1f} us/fwd  {int(r['Calls'])/23:6.1f} calls  {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%  {r['Name'][:100]}")
PY
