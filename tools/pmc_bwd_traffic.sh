#!/bin/bash
# Runs ON the GPU box: only the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh -> gpurun_out/$1/pmc_bench_step.csv
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- $B > $out/pmc_$c.log 2>&1
done
cd $R
{
  echo "# srchash=$(cat snipper_amd/libsnipper_msda.so.srchash)"
  echo "kernel,counter,dispatches,mean_value_KB"
  for c in FETCH_SIZE WRITE_SIZE; do
    f=$(ls $out/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py $f $c snipper:: | head -60
  done
} > $out/pmc_bench_step.csv
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
grep "tile3\|patchbin\|far_kernel" $out/pmc_bench_step.csv
