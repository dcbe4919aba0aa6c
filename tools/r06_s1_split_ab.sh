#!/bin/bash
# Stage-1 train step with the visual tower as two halves on two streams (SF_S1_SPLIT=1, product) against the single-stream tower (SF_S1_SPLIT=0); interleaved, one box
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
run() { env "$@" python bench.py --workload stage1 --batch 2 --steps 10 --warmup 3 --no-cpu-baseline --no-workloads --no-kernel-timing 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['ms_per_step'], 'ms/step', d['value'], 'clips/s')"; }
for rep in 1 2 3; do
  echo "single stream: $(run SF_S1_SPLIT=0)"
  echo "two halves   : $(run SF_S1_SPLIT=1)"
done
