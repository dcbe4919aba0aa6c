#!/bin/bash
# A/B of the two-stream split of the visual tower at the benchmarked batch (16 clips = 224 segments; SF_VIS_SPLIT_MIN=2 SF_VIS_SPLIT_MAX=224 forces it there).  The half-grid
# and lagged variants of round 6 (sf_set_cu_limit, SF_VIS_SPLIT_CUS / _LAG: 3 % SLOWER, profiles/r06_small_m.md) were removed again after the measurement - commit 7ed32d6 has them.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d.get('power',{}); print(d['value'], 'clips/s', d['ms_per_step'], 'ms', p.get('socket_w'), 'W', p.get('sclk_mhz'), 'MHz', p.get('joules_per_clip'), 'J/clip')"; }
for rep in 1 2; do
  echo "default:            $(run SF_VIS_SPLIT_MAX=0)"
  echo "split:              $(run SF_VIS_SPLIT_MIN=2 SF_VIS_SPLIT_MAX=224)"

done
