#!/bin/bash
# Runs ON the GPU box: the tile kernels' walk region by region (default) against level after level (debug bit 128) --
# kernel durations of the encoder backward at sigma 0 / 3 / 8 px (N = 8, bf16 rows, head-major value), same results either way.
cd $GRAFT_REPO_ROOT
for s in 0 3 8; do
  VL=1 bash tools/kstats_t3.sh $s 0 128 0 128
done
