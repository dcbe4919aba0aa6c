#!/bin/bash
# PMC passes (counters only) over the GEMM microbenchmark for one tile config: tools/pmc_pp.sh <outdir> [nseg]   (CFGS=11 SHAPES=qkv,fc1+gelu LIB=...)
OUT=$1; NSEG=${2:-224}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CFGS=${CFGS:-11} KMAJOR=0 ROUNDS=2
[ -n "$LIB" ] && export SYNCHFORMER_HIP_LIB=$LIB
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/tools/bench_gemm.py $NSEG > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('$OUT/*/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('GRBM_GUI_ACTIVE',): n[k] += 1
dur = collections.defaultdict(float); dn = collections.Counter()
for f in glob.glob('$OUT/grbm/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
        dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); dn[k] += 1
for k, c in acc.items():
    if 'gemm' not in k: continue
    nn = max(n[k], 1)
    gui = c['GRBM_GUI_ACTIVE'] / nn
    us = dur[k] / max(dn[k], 1) / 1e3
    wc = c['SQ_WAVE_CYCLES'] or 1
    print(f"{k}: launches {nn} avg {us:.1f} us  clock {gui / us / 1e3:.2f} GHz | MFMA busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * c['GRBM_GUI_ACTIVE']) if c['GRBM_GUI_ACTIVE'] else 0:.1f} % "
          f"| wait_any {100 * c['SQ_WAIT_ANY'] / wc:.1f} % wait_inst {100 * c['SQ_WAIT_INST_ANY'] / wc:.1f} % active {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.1f} % "
          f"| LDS conflict {100 * c['SQ_LDS_BANK_CONFLICT'] / (c['SQ_LDS_IDX_ACTIVE'] or 1):.1f} % | fetch x2 {c['FETCH_SIZE'] * 2 / 1024 / nn:.0f} MiB/launch")
PY
