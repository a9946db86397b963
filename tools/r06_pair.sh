#!/bin/bash
# Runs ON the GPU box: forward kernel with both x-neighbour taps per load instruction (head-major bf16 value) -- parity tests, then
# A/B of the step and of the launch alone (SNIPPER_MSDA_FWD_PAIR=0 = one tap per instruction).
tag=${1:-r06j}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python3 -m pytest tests/test_timed_path_gpu.py tests/test_owner_gpu.py tests/test_msda_gpu.py tests/test_encoder_native_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
for rep in 1 2 3; do
  for pr in 1 0; do
    SNIPPER_MSDA_FWD_PAIR=$pr python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_pair${pr}_$rep.json 2> $out/bench_pair${pr}_$rep.err
    python3 - $out/bench_pair${pr}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "fwd ms/step", {k: v for k, v in d["msda_launch_ms_per_step"].items() if k.startswith("fwd")}, "loss", d["final_loss"])
PY
  done
done
SNIPPER_MSDA_FWD_PAIR=1 bash tools/bench_kstats.sh ${tag}_pair1 > $out/kstats_pair1.txt 2>&1
SNIPPER_MSDA_FWD_PAIR=0 bash tools/bench_kstats.sh ${tag}_pair0 > $out/kstats_pair0.txt 2>&1
grep -E "msda_fwd|total kernel" $out/kstats_pair1.txt $out/kstats_pair0.txt
