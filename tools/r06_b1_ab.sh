#!/bin/bash
# A/B of the small-batch visual split (SF_VIS_SPLIT_MAX) on the one-clip and two-clip forwards, interleaved repetitions on one box (run on the GPU box)
for rep in 1 2 3; do
  for split in 0 28; do
    for B in 1 2; do
      SF_VIS_SPLIT_MAX=$split python tools/b1_forward.py 30 $B 2>&1 | grep clip
    done
  done
done
