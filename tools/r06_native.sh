#!/bin/bash
# Runs ON the GPU box: native decoder layer -- tests, then A/B of the step (native on / off) with the host-issue time.
tag=${1:-r06d}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_decoder_native_gpu.py tests/test_decoder_chain_gpu.py tests/test_module_gpu.py tests/test_config2_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -30 $out/tests.txt
for rep in 1 2; do
  for nat in 1 0; do
    SNIPPER_DEC_NATIVE=$nat python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_native${nat}_$rep.json 2> $out/bench_native${nat}_$rep.err
    python3 - $out/bench_native${nat}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "loss", d["final_loss"], "dec", d["msda"]["decoder_module_fwd_bwd_ms"])
PY
  done
done
