#!/bin/bash
tag=${1:-r06q}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for cfg in "X=1" "SNIPPER_WGRAD_RING_TILES=16" "SNIPPER_WGRAD_RING_TILES=32" "SNIPPER_WGRAD_RING=0" "SNIPPER_WGRAD_S=16" "SNIPPER_WGRAD_S=24" "SNIPPER_WGRAD_S=48" "SNIPPER_WGRAD_S=64"; do
  env $cfg python3 tools/wgrad_sweep.py 2>&1 | grep -v amdgpu.ids
done | tee $out/wgrad_sweep.jsonl
