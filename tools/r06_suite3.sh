#!/bin/bash
tag=${1:-r06x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for i in 1 2 3; do timeout 600 python3 -m pytest tests/test_encoder_native_gpu.py -x -q -s 2>&1 | grep -E "native vs per-module|passed|failed"; done | tee $out/encoder_native_repeats.txt
timeout 1500 python3 -m pytest tests -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
tail -4 $out/gputests.txt
