#!/bin/bash
tag=${1:-r06t}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for rep in 1 2; do
  for m in 1 0; do
    SNIPPER_LINEAR_PATCH=$m python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_lp${m}_$rep.json 2> $out/bench_lp${m}_$rep.err
    python3 - $out/bench_lp${m}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "dense", d["roofline_dense"]["ms_per_step"], d["roofline_dense"]["frac"], "launches", d.get("launches_per_step"), "issue", d.get("host_issue_ms"), "loss", d["final_loss"])
PY
  done
done | tee $out/step_ab.txt
