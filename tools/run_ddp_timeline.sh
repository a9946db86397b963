#!/bin/bash
# Runs ON the GPU box: kernel trace of the staged all-reduce path on ONE GPU (1-rank RCCL group) + its timeline summary,
# and the same command without the forced path for the +ms/step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ddp_trace
SNIPPER_FORCE_DDP=1 SNIPPER_SYNC_FORCE=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 8 --warmup 6 --no-extras --no-cpu-baseline > $out.json 2> $out.err
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 $R/tools/ddp_timeline.py $f > $R/gpurun_out/ddp_timeline.txt 2>&1
rm -rf $out
cd $R
SNIPPER_FORCE_DDP=1 SNIPPER_SYNC_FORCE=1 python3 bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced staged all-reduce (1-rank RCCL): ms_per_step', d['ms_per_step'])" >> gpurun_out/ddp_timeline.txt
python3 bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain 1-GPU step: ms_per_step', d['ms_per_step'])" >> gpurun_out/ddp_timeline.txt
cat gpurun_out/ddp_timeline.txt
