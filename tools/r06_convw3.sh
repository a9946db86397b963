#!/bin/bash
# Runs ON the GPU box: step-level A/B of the patch-resident 3x3 weight gradient (SNIPPER_WGRAD_CONV_PATCH=0 = conv mode of the
# split-reduction kernel), backbone / dense tests.
tag=${1:-r06p}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python3 -m pytest tests/test_dense_gpu.py tests/test_config2_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
for rep in 1 2 3; do
  for pr in 1 0; do
    SNIPPER_WGRAD_CONV_PATCH=$pr python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_patch${pr}_$rep.json 2> $out/bench_patch${pr}_$rep.err
    python3 - $out/bench_patch${pr}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "dense", d["roofline_dense"]["ms_per_step"], d["roofline_dense"]["frac"], "loss", d["final_loss"])
PY
  done
done
