#!/bin/bash
# clips per GPU per step: bench.py --batch B for B in 1 2 4 8 16 32 (one MI355X); prints one line per B
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for b in ${BATCHES:-1 2 4 8 16 32}; do
  python $R/bench.py --batch $b --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-workloads 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1]); print(f'| {d[\"config\"][\"clips_per_gpu\"]} | {d[\"value\"]} | {d[\"ms_per_step\"]} |')"
done
