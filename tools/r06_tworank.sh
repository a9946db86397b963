#!/bin/bash
# Runs ON the GPU box: the N > 1 bench path with TWO ranks sharing the one GPU (gloo carries the collectives on device tensors; RCCL
# refuses two ranks per device).  A path test -- barriers, staged all-reduce hooks, gathers, cross-rank parameter check -- not a
# measurement: both ranks' kernels interleave on one device.
tag=${1:-r06k}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
export SNIPPER_DIST_BACKEND=gloo SNIPPER_SHARE_GPU=1
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 4 > $out/bench_two_ranks_one_gpu.json 2> $out/bench_two_ranks_one_gpu.err
echo "rc=$?"
tail -5 $out/bench_two_ranks_one_gpu.err
python3 - $out/bench_two_ranks_one_gpu.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "final_loss", "host_issue_ms")})
print(json.dumps(d["distributed"], indent=1))
PY
