#!/bin/bash
# Runs ON the GPU box: per-kernel statistics of the default bench step (rocprofv3 kernel trace), top kernels printed.
# usage: tools/bench_kstats.sh <tag> [bench args]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 8 --no-extras --no-cpu-baseline "$@" > $out.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.csv
t=$(ls $out/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py $t > $GRAFT_REPO_ROOT/gpurun_out/kseq_$tag.txt 2>&1
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms per step: {tot / 20 / 1e6:.2f}  (20 steps incl. warm-up)")
for r in rows[:28]:
    n = r["Name"]
    short = n.split("snipper::")[1].split("(")[0] if "snipper::" in n else n[:60]
    print(f"{short:62s} calls/step {int(r['Calls']) / 20:6.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / 20 / 1e6:6.3f}")
PY
rm -rf $out
