#!/bin/bash
# Runs ON the GPU box: the full GPU test suite, the default bench line, untuned-host and forced-DDP lines (VERDICT r05 #2's criteria).
tag=${1:-r06g}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
tail -4 $out/gputests.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python3 bench.py --steps 20 --warmup 5 --pin-cores 0 --gc-every 0 --no-cpu-baseline --no-locality-sweep > $out/bench_host_untuned.json 2> $out/bench_host_untuned.err
SNIPPER_FORCE_DDP=1 SNIPPER_SYNC_FORCE=1 python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-locality-sweep > $out/bench_forced_ddp.json 2> $out/bench_forced_ddp.err
SNIPPER_ISSUE_TIME=5 SNIPPER_REGION_EVENTS=12 python3 bench.py --steps 10 --warmup 8 --no-extras --no-cpu-baseline > /dev/null 2> $out/region_events.err
grep -E "region_events|issue " $out/region_events.err > $out/region_events.txt
for f in default host_untuned forced_ddp; do python3 - $out/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], "issue", d.get("host_issue_ms"), "launches", d.get("launches_per_step"), "<20us", d.get("kernels_under_20us_ms"), "roofline", (d.get("roofline") or {}).get("frac"), "s3", d.get("ms_per_step_at_offset_sigma_3px"), "dec", (d.get("msda") or {}).get("decoder_module_fwd_bwd_ms"))
PY
done
cat $out/region_events.txt | cut -c1-300
