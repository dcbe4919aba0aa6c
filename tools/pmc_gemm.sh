#!/bin/bash
# PMC passes over the GEMM microbenchmark (counters only: no --sys-trace etc.).  Usage: tools/pmc_gemm.sh <outdir> <nseg>
OUT=$1; NSEG=${2:-27}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/tools/bench_gemm.py $NSEG > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE GRBM_GUI_ACTIVE
ls $OUT
