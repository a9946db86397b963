#!/bin/bash
# Runs ON the GPU box: the four parity classes of a stride-2 3x3 data gradient as one launch -- parity tests, per-op times, step A/B.
tag=${1:-r06s}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_dense_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
for m in 1 0; do echo "== SNIPPER_DGRAD2_MERGE=$m"; SNIPPER_DGRAD2_MERGE=$m python3 tools/convbench.py 2>&1 | grep -i "dgrad_s2\|s2"; done | tee $out/convbench_dgrad2.txt
for rep in 1 2 3; do
  for m in 1 0; do
    SNIPPER_DGRAD2_MERGE=$m python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_merge${m}_$rep.json 2> $out/bench_merge${m}_$rep.err
    python3 - $out/bench_merge${m}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "dense", d["roofline_dense"]["ms_per_step"], d["roofline_dense"]["frac"], "launches", d.get("launches_per_step"), "loss", d["final_loss"])
PY
  done
done | tee $out/step_ab.txt
