#!/bin/bash
# VERDICT r5 item 5 (run on the GPU box): the two fused attention launches - fabric bytes by head-pair sweep (SF_QT2_PAIR_CHUNK / SF_QS_PAIR_CHUNK = 6 | 3 | 2), LDS bank
# conflicts with the hand-over stores made conflict-free (ablation library tools/ab_build/libsf_1.so built by `SRC=sf_qkv_time2 tools/ab_pp.sh "-DQT2_ABL_HALF=1"`),
# and what the sweep costs end to end in clips/s and J/clip.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
out=$R/gpurun_out/r06/fabric; mkdir -p $out
export PMC_GROUPS="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT;FETCH_SIZE;WRITE_SIZE GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for hc in 6 3 2; do
  echo "=== qkv_time2, SF_QT2_PAIR_CHUNK=$hc"; SF_QT2_PAIR_CHUNK=$hc python $R/tools/bench_qkv_time2.py 224 2>&1 | grep n_seg
  SF_QT2_PAIR_CHUNK=$hc bash $R/tools/pmc_kernels.sh qkv_time2_attn python $R/tools/bench_qkv_time2.py 224
  echo "=== qkv_space, SF_QS_PAIR_CHUNK=$hc"; SF_QS_PAIR_CHUNK=$hc python $R/tools/bench_qkv_space.py 224 2>&1 | grep n_seg
  SF_QS_PAIR_CHUNK=$hc bash $R/tools/pmc_kernels.sh qkv_space_attn python $R/tools/bench_qkv_space.py 224
done > $out/sweep.txt 2>&1
if [ -f $R/tools/ab_build/libsf_1.so ]; then
  { echo "=== qkv_time2 with conflict-free hand-over stores (QT2_ABL_HALF=1, wrong results)"
    for rep in 1 2 3; do
      SYNCHFORMER_HIP_LIB=$R/tools/ab_build/libsf_1.so python $R/tools/bench_qkv_time2.py 224 2>&1 | grep n_seg
      python $R/tools/bench_qkv_time2.py 224 2>&1 | grep n_seg
    done
    PMC_GROUPS="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" SYNCHFORMER_HIP_LIB=$R/tools/ab_build/libsf_1.so bash $R/tools/pmc_kernels.sh qkv_time2_attn python $R/tools/bench_qkv_time2.py 224
  } > $out/half.txt 2>&1
fi
cd $R
for rep in 1 2 3; do
  for hc in 6 3; do
    echo "=== e2e SF_QT2_PAIR_CHUNK=$hc SF_QS_PAIR_CHUNK=$hc (rep $rep)"
    SF_QT2_PAIR_CHUNK=$hc SF_QS_PAIR_CHUNK=$hc python bench.py --steps 10 --warmup 3 --no-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d.get('power',{}); print(d['value'], 'clips/s', d['ms_per_step'], 'ms', p.get('socket_w'), 'W', p.get('sclk_mhz'), 'MHz', p.get('joules_per_clip'), 'J/clip')"
  done
done > $out/e2e.txt 2>&1
tail -n 30 $out/e2e.txt
