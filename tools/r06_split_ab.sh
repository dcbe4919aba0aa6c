#!/bin/bash
# A/B of the two-stream split of the visual tower at the benchmarked batch (16 clips = 224 segments): plain | half-grid launches | half-grid + the second half lagging
# by k launch groups of a block (run on the GPU box; interleaved repetitions).  python bench.py prints clips/s, W, J/clip.
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d.get('power',{}); print(d['value'], 'clips/s', d['ms_per_step'], 'ms', p.get('socket_w'), 'W', p.get('sclk_mhz'), 'MHz', p.get('joules_per_clip'), 'J/clip')"; }
for rep in 1 2; do
  echo "default:            $(run SF_VIS_SPLIT_MAX=0)"
  echo "split:              $(run SF_VIS_SPLIT_MAX=224)"
  echo "split cus128:       $(run SF_VIS_SPLIT_MAX=224 SF_VIS_SPLIT_CUS=128)"
  echo "split cus128 lag3:  $(run SF_VIS_SPLIT_MAX=224 SF_VIS_SPLIT_CUS=128 SF_VIS_SPLIT_LAG=3)"
  echo "split cus128 lag5:  $(run SF_VIS_SPLIT_MAX=224 SF_VIS_SPLIT_CUS=128 SF_VIS_SPLIT_LAG=5)"
  echo "split lag3:         $(run SF_VIS_SPLIT_MAX=224 SF_VIS_SPLIT_LAG=3)"
  echo "split cus192 lag3:  $(run SF_VIS_SPLIT_MAX=224 SF_VIS_SPLIT_CUS=192 SF_VIS_SPLIT_LAG=3)"
done
