#!/bin/bash
# Runs ON the GPU box: attention for 257..384 queries + small-LN column pass -- tests, the forecast bench line, kernel table.
tag=${1:-r06h}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1200 python3 -m pytest tests/test_module_gpu.py tests/test_decoder_chain_gpu.py tests/test_decoder_native_gpu.py tests/test_forecast_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -6 $out/tests.txt
python3 bench.py --future-frames 2 --no-cpu-baseline > $out/bench_forecast.json 2> $out/bench_forecast.err
python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_default.json 2> $out/bench_default.err
for f in forecast default; do python3 - $out/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "dec", (d.get("msda") or {}).get("decoder_module_fwd_bwd_ms"))
PY
done
bash tools/bench_kstats.sh ${tag}_forecast --future-frames 2 > $out/kstats_forecast.txt 2>&1
bash tools/bench_kstats.sh ${tag}_default > $out/kstats_default.txt 2>&1
grep -E "small_attn|small_ln|total kernel|attn_fwd|bwd_kernel_d" $out/kstats_forecast.txt $out/kstats_default.txt
