#!/bin/bash
# Runs ON the GPU box: same-box A/B of two builds of the library (SNIPPER_MSDA_LIB selects the file): step time and one kernel's average.
# usage: tools/r06_ab_lib.sh <tag> <kernel name pattern>
tag=${1:-r06z}; pat=${2:-small_gemm_batch}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
for rep in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then export SNIPPER_MSDA_LIB=libsnipper_old.so; else unset SNIPPER_MSDA_LIB; fi
    python3 bench.py --no-cpu-baseline --no-locality-sweep > $out/bench_${lib}_$rep.json 2> $out/bench_${lib}_$rep.err
    python3 - $out/bench_${lib}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["ms_per_step"], "<20us", d.get("kernels_under_20us_ms"), "dec", d["msda"].get("decoder_module_fwd_bwd_ms"), "loss", d["final_loss"])
PY
  done
done
for lib in new old; do
  if [ $lib = old ]; then export SNIPPER_MSDA_LIB=libsnipper_old.so; else unset SNIPPER_MSDA_LIB; fi
  bash tools/bench_kstats.sh ${tag}_$lib > $out/kstats_$lib.txt 2>&1
  echo "$lib: $(grep -E "total kernel" $out/kstats_$lib.txt) | $(grep -E "$pat" $out/kstats_$lib.txt)"
done
