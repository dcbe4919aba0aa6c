run() { echo -n "$1: "; env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-workloads 2>&1 | tail -1 | cut -c50-66; }
for r in 1 2; do
  run "X=0"; run "SF_FUSE_LN_FC2=0"; run "SF_FUSE_LN=0"; run "SF_FUSE_TIME=0"; run "SF_AUDIO_SIDE_STREAM=0"; run "SF_GEMM_NCHUNK=0"; run "SF_GEMM_NCHUNK=3"; run "SF_GEMM_NCHUNK=6"; run "SF_CLS_FUSION=none"
done
