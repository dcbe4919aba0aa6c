#!/bin/bash
# one-clip forward under each schedule switch of the engine (is any un-fused launch faster at M = 21,966 rows?); run on the GPU box
for rep in 1 2; do
  for kv in "" "SF_FUSE_TIME2=0" "SF_FUSE_SPACE=0" "SF_FUSE_LN=0" "SF_FUSE_LN_FC2=0" "SF_FUSE_TIME=0" "SF_FUSE_TIME2=0 SF_FUSE_SPACE=0" "SF_PE_TOKENS=0" "SF_AUDIO_SIDE_STREAM=0"; do
    echo "$(env $kv SF_VIS_SPLIT_MAX=0 python tools/b1_forward.py 30 1 2>&1 | grep clip)   [$kv]"
  done
done
