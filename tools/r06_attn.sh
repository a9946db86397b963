#!/bin/bash
# Runs ON the GPU box: small-attention backward with 32-row workgroups -- tests, then A/B of the step (rows forced 16 / 32).
tag=${1:-r06c}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_decoder_chain_gpu.py tests/test_module_gpu.py -x -q -k "attention or chain or transformer" > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
for rep in 1 2; do
  for rows in 32 16; do
    SNIPPER_SMALL_ATTN_ROWS=$rows python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_rows${rows}_$rep.json 2> $out/bench_rows${rows}_$rep.err
    python3 - $out/bench_rows${rows}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "loss", d["final_loss"], "dec", d["msda"]["decoder_module_fwd_bwd_ms"])
PY
  done
done
bash tools/bench_kstats.sh ${tag} > $out/kstats.txt 2>&1
grep -E "small_attn|total kernel" $out/kstats.txt
