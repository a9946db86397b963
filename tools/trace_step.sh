#!/bin/bash
# Runs ON the GPU box: one kernel trace of the default bench step -> gap analysis + launch-ordered listing of the last step.
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $R/bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $out/trace.log 2>&1
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
cd $R
python3 tools/gap_analysis.py $f --delim adamw_clip_kernel --steps 3 --top 40 > $out/gaps.txt 2>&1
python3 tools/step_sequence.py $f > $out/sequence.txt 2>&1
rm -rf $out/trace
