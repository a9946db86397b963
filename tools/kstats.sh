#!/bin/bash
# usage: tools/kstats.sh <tag> <opbench args...>   -- per-kernel average durations (us) of one opbench run under rocprofv3
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/opbench.py "$@" > $out.log 2>&1
f=$(ls -t $out/*/*kernel_stats.csv | head -1)
python3 - "$f" "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "snipper" in n:
        short = n.split("snipper::")[1].split("(")[0]
        print(f"{sys.argv[2]:10s} {short:45s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
