#!/bin/bash
# Runs ON the GPU box: 3x3 weight gradient with the input patch resident in LDS -- parity, then kernel + second pass per shape
# for the patch kernel (2 / 4 co blocks per wave, grid sizes) against the conv mode of the split-reduction kernel.
tag=${1:-r06m}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_dense_gpu.py -x -q -k "conv3x3_weight_gradient or whole_backbone" > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -4 $out/tests.txt
for cfg in "SNIPPER_WGRAD_CONV_PATCH=0" "SNIPPER_WGRAD_CONV_CB=2" "SNIPPER_WGRAD_CONV_CB=4" "SNIPPER_WGRAD_CONV_CB=2 SNIPPER_WGRAD_CONV_WGS=256" \
           "SNIPPER_WGRAD_CONV_CB=4 SNIPPER_WGRAD_CONV_WGS=256" "SNIPPER_WGRAD_CONV_CB=2 SNIPPER_WGRAD_CONV_WGS=768"; do
  echo "== $cfg"
  env $cfg python3 tools/convwgradbench.py 2>&1 | grep -v amdgpu.ids | head -3
done | tee $out/convwgradbench_ab.txt
