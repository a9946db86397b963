#!/bin/bash
# Runs ON the GPU box (gpurun): collects the round's measurement evidence into gpurun_out/$1/
#   PMC FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the bench command (separate passes, kernel-trace only),
#   the copy-kernel HBM ceiling, the op-level locality sweep, and the fp32 bench line.
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extras > $out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-extras > $out/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
{
  echo "kernel,counter,dispatches,mean_value_KB"
  for c in FETCH_SIZE WRITE_SIZE; do
    f=$(ls $out/pmc_$c/*/*counter_collection.csv | head -1)
    python3 tools/pmc_summary.py $f $c snipper:: | head -14
  done
} > $out/pmc_bench_step.csv
{
  echo "kernel,counter,dispatches,mean_value"
  f=$(ls $out/pmc_mfma/*/*counter_collection.csv | head -1)
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    python3 tools/pmc_summary.py $f $c linear_bf16 wgrad_bf16_kernel conv3x3 stem7x7 | head -9
  done
} > $out/pmc_mfma_busy.csv
python3 tools/copybench.py > $out/copybench.json 2> $out/copybench.err
python3 tools/opbench.py --N 8 --cases enc_local --dtypes float32 --skip-torch --rows-bf16 1 --sigma 1 3 8 --far 0 0.1 0.5 --iters 10 > $out/opbench_locality.jsonl 2> $out/opbench.err
python3 tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 1 3 8 --far 0 0.5 --iters 10 > $out/opbench_locality_bf16_value.jsonl 2>> $out/opbench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 8 --no-extras > $out/kstats.log 2>&1)
python3 bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err
python3 bench.py --steps 20 --warmup 5 > $out/bench_bf16.json 2> $out/bench_bf16.err
tail -c 600 $out/bench_bf16.json; cat $out/copybench.json; head -3 $out/pmc_mfma_busy.csv; grep -c case $out/opbench_locality.jsonl
