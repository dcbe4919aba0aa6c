#!/bin/bash
# HBM-side fetch of the GEMM shapes under different column-chunk sweeps of the persistent kernel (experiment)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${CHUNKS:-0 2 3 5}; do
  rm -rf /tmp/pmc_c$c
  SF_GEMM_NCHUNK=$c CFGS=7 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_c$c -o f -- python $R/tools/bench_gemm.py 224 > /tmp/pmc_c$c.log 2>&1
  f=$(find /tmp/pmc_c$c -name '*counter_collection.csv' | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
tot=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][-40:]
    if r['Counter_Name']=='FETCH_SIZE':
        tot[k]+=float(r['Counter_Value']); n[k]+=1
for k in tot:
    if 'persistent' in k: print('chunk', sys.argv[2], k, 'launches', n[k], 'fetch x2 MiB/launch', round(tot[k]*2/1024/n[k]))
PY
  grep "qkv\|fc1\|fc2" /tmp/pmc_c$c.log | head -3
done
