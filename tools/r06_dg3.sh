#!/bin/bash
tag=${1:-r06w}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python3 -m pytest tests/test_dense_gpu.py -x -q > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
python3 tools/convbench.py 2>&1 | grep -i "s2" | tee $out/convbench_dgrad2.txt
bash tools/bench_kstats.sh $tag > $out/kstats.txt 2>&1
grep -E "total kernel|conv3x3_ring_kernel<false>|conv3x3_bf16_kernel<false" $out/kstats.txt
for rep in 1 2; do
python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_$rep.json 2> $out/bench_$rep.err
python3 - $out/bench_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "dense", d["roofline_dense"]["ms_per_step"], d["roofline_dense"]["frac"], "launches", d.get("launches_per_step"), "loss", d["final_loss"])
PY
done
