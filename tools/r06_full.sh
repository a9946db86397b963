#!/bin/bash
# Runs ON the GPU box: full GPU test suite + kernel tables with the native decoder layer on / off (launch-count diff).
tag=${1:-r06e}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/gputests.txt 2>&1; echo "pytest rc=$?" >> $out/gputests.txt
tail -5 $out/gputests.txt
SNIPPER_DEC_NATIVE=1 bash tools/bench_kstats.sh ${tag}_native1 > $out/kstats_native1.txt 2>&1
SNIPPER_DEC_NATIVE=0 bash tools/bench_kstats.sh ${tag}_native0 > $out/kstats_native0.txt 2>&1
python3 - $R/gpurun_out/kstats_${tag}_native1.csv $R/gpurun_out/kstats_${tag}_native0.csv <<'PY'
import csv, sys
a = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(sys.argv[2]))}
for k in sorted(set(a) | set(b)):
    if a.get(k, 0) != b.get(k, 0):
        print(f"{(a.get(k, 0) - b.get(k, 0)) / 20:+7.2f} per step  native {a.get(k, 0) / 20:7.2f}  chain {b.get(k, 0) / 20:7.2f}  {k[:110]}")
PY
