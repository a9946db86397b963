#!/bin/bash
# usage: [EDGES="16 8 4"] [VL=1 (head-major value layout)] tools/kstats_t3.sh <sigma> <debug flags...>  -- per-kernel average durations of the encoder backward
# (N = 8, bf16) with the matrix-pipe tile kernel's phases compiled out at run time (WRONG results; timing only)
sigma=$1; shift
export SNIPPER_MSDA_ALLOW_DEBUG=1
for dbg in "$@"; do
  extra=""; [ "$dbg" != "0" ] && extra="--debug $dbg"
  tag=t3_s${sigma}_d${dbg}_e$(echo $EDGES | tr ' ' '-')
  bash $GRAFT_REPO_ROOT/tools/kstats.sh $tag --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma $sigma --grid 1 --iters 5 --tile-kernel ${TK:-2} --value-layout ${VL:-0} $extra ${EDGES:+--edges $EDGES} | grep -i "tile\|patchbin"
done
