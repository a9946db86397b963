#!/bin/bash
# Runs ON the GPU box: the round's final evidence -- full GPU suite, tools/collect_profiles.sh (PMC passes, kernel table, dense table,
# fp32 / forced-DDP / untuned-host lines), the forecast and 540x960 lines, a plain line with the forced-DDP run's step counts.
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
tail -4 $out/gputests.txt
bash tools/collect_profiles.sh $tag > $out/collect.log 2>&1
# the default line once more with this library's counter profile in place (bench.py quotes roofline.traffic from profiles/ only when
# the profile's source hash equals the loaded library's)
cp $out/pmc_bench_step.csv profiles/r06_pmc_bench_step.csv
python3 bench.py > $out/bench_bf16_default.json 2> $out/bench_bf16_default.err
python3 bench.py --future-frames 2 --no-cpu-baseline > $out/bench_forecast.json 2> $out/bench_forecast.err
python3 bench.py --height 540 --width 960 --no-cpu-baseline > $out/bench_540x960.json 2> $out/bench_540x960.err
python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-locality-sweep > $out/bench_plain_20_8.json 2> $out/bench_plain_20_8.err
SNIPPER_ENC_NATIVE=0 SNIPPER_DEC_NATIVE=0 SNIPPER_DEC_CHAIN=0 python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_per_module.json 2> $out/bench_per_module.err
bash tools/bench_kstats.sh ${tag}_forecast --future-frames 2 > $out/kstats_forecast.txt 2>&1
bash tools/bench_kstats.sh ${tag}_540x960 --height 540 --width 960 > $out/kstats_540x960.txt 2>&1
SNIPPER_ISSUE_TIME=5 SNIPPER_REGION_EVENTS=12 python3 bench.py --steps 10 --warmup 8 --no-extras --no-cpu-baseline > /dev/null 2> $out/region_events.err
grep -E "region_events|issue " $out/region_events.err > $out/region_events.txt
for f in bf16_default forecast 540x960 plain_20_8 forced_ddp host_untuned per_module fp32; do python3 - $out/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), "loss", d["final_loss"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
