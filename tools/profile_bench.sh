#!/bin/bash
# rocprofv3 profiles of the headline benchmark (run on the GPU box through gpurun).
#   tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_*}/...
# Kernel-trace/stats and PMC counters are collected in SEPARATE runs (gpurun refuses combined runs).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the configuration of bench.py's roofline pass: the two towers serialised on one stream (kernel durations are not stretched by co-running kernels)
export SF_AUDIO_SIDE_STREAM=0
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE GRBM_GUI_ACTIVE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o $name -- $BENCH > $OUT/pmc_$name.log 2>&1
done
grep -h '"metric"' $OUT/stats.log | tail -1
ls $OUT $OUT/stats
