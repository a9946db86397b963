#!/bin/bash
tag=${1:-r06u}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
tail -4 $out/gputests.txt
python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python3 - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "dense", d["roofline_dense"]["ms_per_step"], d["roofline_dense"]["frac"], "launches", d.get("launches_per_step"), "issue", d.get("host_issue_ms"), "sigma3", d.get("ms_per_step_at_offset_sigma_3px"), "loss", d["final_loss"])
PY
