#!/bin/bash
# GPU box: per variant built by tools/ab_pp.sh - FETCH_SIZE (x2) and L2 hit rate of the config-11 GEMM shapes (one --pmc pass each).  SHAPES=qkv,fc1+gelu
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
export CFGS=11 KMAJOR=0 ROUNDS=1 ITERS=4 SHAPES=${SHAPES:-qkv,fc1+gelu}
for so in $(ls $R/tools/ab_build/libsf_*.so | sort -V); do
  i=$(basename $so .so | sed s/libsf_//)
  echo "=== variant $i: '$(cat $R/tools/ab_build/flags_$i.txt)'"
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/abf
    SYNCHFORMER_HIP_LIB=$so rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/abf -o f -- python $R/tools/bench_gemm.py ${SEGS:-224} > /tmp/abf.log 2>&1
    f=$(find /tmp/abf -name '*counter_collection.csv' | head -1)
    python - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:64]
    if 'gemm' not in k: continue
    tot[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
ks = sorted({k for k, _ in tot})
for k in ks:
    if (k, 'FETCH_SIZE') in tot: print(f'  {k}: fetch x2 {tot[(k, "FETCH_SIZE")] * 2 / 1024 / n[(k, "FETCH_SIZE")]:.0f} MiB/launch')
    if (k, 'TCC_HIT_sum') in tot: print(f'  {k}: L2 hit {100 * tot[(k, "TCC_HIT_sum")] / (tot[(k, "TCC_HIT_sum")] + tot[(k, "TCC_MISS_sum")]):.1f} %')
PY
  done
done
