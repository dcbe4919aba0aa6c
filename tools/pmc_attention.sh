#!/bin/bash
# SQ counters of the attention kernels on the model's shapes (one --pmc pass per counter group; run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); rm -rf /tmp/pa$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pa$i -o f -- python $R/tools/bench_attention.py ${1:-224} > /tmp/pa$i.log 2>&1
  f=$(find /tmp/pa$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
tot=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][-28:]
    if 'attn' not in k: continue
    tot[(k, r['Counter_Name'])]+=float(r['Counter_Value']); n[(k, r['Counter_Name'])]+=1
for (k, c) in sorted(tot): print(f'{k:30s} {c:28s} per launch {tot[(k,c)]/n[(k,c)]:.4g}')
PY
done
