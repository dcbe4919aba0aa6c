#!/bin/bash
# SQ / memory counters per kernel for any command (one --pmc pass per counter group; run on the GPU box).
#   tools/pmc_kernels.sh <kernel-name-substring> <command ...>
cd /tmp && export TMPDIR=/tmp
FILT=$1; shift
i=0
#   PMC_GROUPS="FETCH_SIZE;WRITE_SIZE GRBM_GUI_ACTIVE" restricts the passes (';'-separated counter groups)
DEFAULT_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY;FETCH_SIZE;WRITE_SIZE GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pk$i -o f -- "$@" > /tmp/pk$i.log 2>&1
  f=$(find /tmp/pk$i -name '*counter_collection.csv' | head -1)
  python - "$f" "$FILT" <<'PY'
import csv, sys, collections
tot=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    if sys.argv[2] not in k: continue
    k=k[-44:]
    tot[(k, r['Counter_Name'])]+=float(r['Counter_Value']); n[(k, r['Counter_Name'])]+=1
for (k, c) in sorted(tot): print(f'{k:46s} {c:28s} launches {n[(k,c)]:4d} per launch {tot[(k,c)]/n[(k,c)]:.4g}')
PY
done
