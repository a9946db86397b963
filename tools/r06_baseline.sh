#!/bin/bash
# Runs ON the GPU box: round-6 baseline -- GPU tests, the default bench line, the two configurations nobody had timed
# (BASELINE configs[4]: --future-frames 2; the README's 540x960 input), their kernel tables, region events.
# usage: tools/r06_baseline.sh <tag> [skip-tests]
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
if [ -z "$2" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
  tail -3 $out/gputests.txt
fi
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python3 bench.py --future-frames 2 --no-cpu-baseline > $out/bench_forecast.json 2> $out/bench_forecast.err; echo "forecast rc=$?"
python3 bench.py --height 540 --width 960 --no-cpu-baseline > $out/bench_540x960.json 2> $out/bench_540x960.err; echo "540 rc=$?"
SNIPPER_REGION_EVENTS=12 SNIPPER_ISSUE_TIME=5 python3 bench.py --steps 10 --warmup 8 --no-extras --no-cpu-baseline > /dev/null 2> $out/region_events.err
grep -E "region_events|issue " $out/region_events.err > $out/region_events.txt
bash tools/bench_kstats.sh ${tag}_default > $out/kstats_default.txt 2>&1
bash tools/bench_kstats.sh ${tag}_forecast --future-frames 2 > $out/kstats_forecast.txt 2>&1
bash tools/bench_kstats.sh ${tag}_540x960 --height 540 --width 960 > $out/kstats_540x960.txt 2>&1
for f in $out/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"),
          "<20us ms", d.get("kernels_under_20us_ms"), "roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_launch_ms"),
          "dense", (d.get("roofline_dense") or {}).get("ms_per_step"), "msda", {k: v for k, v in (d.get("msda") or {}).items() if "ms" in k})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
