#!/bin/bash
# Runs ON the GPU box: native encoder layer -- tests, then A/B of the step (native on / off) with the host-issue time.
tag=${1:-r06f}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_encoder_native_gpu.py tests/test_decoder_native_gpu.py tests/test_config2_gpu.py tests/test_training_parity_gpu.py tests/test_timed_path_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -30 $out/tests.txt
for rep in 1 2; do
  for nat in 1 0; do
    SNIPPER_ENC_NATIVE=$nat python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_encnative${nat}_$rep.json 2> $out/bench_encnative${nat}_$rep.err
    python3 - $out/bench_encnative${nat}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "loss", d["final_loss"], "enc", d["msda"]["encoder_module_fwd_bwd_ms"], "roofline", d["roofline"]["frac"])
PY
  done
done
