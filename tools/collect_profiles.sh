#!/bin/bash
# Runs ON the GPU box (gpurun): collects the round's measurement evidence into gpurun_out/$1/ (copy what is to be judged
# into profiles/).  PMC passes are separate, kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline"
pass() {   # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$name -- $B > $out/pmc_$name.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum
cd $R
sumfile() { ls $out/pmc_$1/*/*counter_collection.csv 2>/dev/null | head -1; }
{
  echo "# srchash=$(cat snipper_amd/libsnipper_msda.so.srchash)"
  echo "kernel,counter,dispatches,mean_value_KB"
  for c in FETCH_SIZE; do f=$(sumfile fetch); [ -n "$f" ] && python3 tools/pmc_summary.py $f $c snipper:: | head -60; done
  for c in WRITE_SIZE; do f=$(sumfile write); [ -n "$f" ] && python3 tools/pmc_summary.py $f $c snipper:: | head -60; done
} > $out/pmc_bench_step.csv
{
  echo "kernel,counter,dispatches,mean_value"
  f=$(sumfile mfma)
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    [ -n "$f" ] && python3 tools/pmc_summary.py $f $c wres_gemm wgrad_ring wgrad_bf16_kernel linear_bf16 conv3x3 stem7x7 | head -14
  done
} > $out/pmc_mfma_busy.csv
{
  echo "kernel,counter,dispatches,mean_value"
  f=$(sumfile lds)
  for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; do
    [ -n "$f" ] && python3 tools/pmc_summary.py $f $c msda_bwd_d48_tile3 msda_bwd_d48_patchbin msda_fwd_d48 wres_gemm wgrad_ring
  done
  f=$(sumfile tcc)
  for c in TCC_HIT_sum TCC_MISS_sum; do
    [ -n "$f" ] && python3 tools/pmc_summary.py $f $c msda_bwd_d48_tile3 msda_bwd_d48_patchbin msda_fwd_d48 wres_gemm wgrad_ring wgrad_bf16_kernel
  done
} > $out/pmc_lds_tcc.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_lds $out/pmc_tcc
python3 tools/copybench.py > $out/copybench.json 2> $out/copybench.err
python3 tools/graphbench.py > $out/graphbench.jsonl 2> $out/graphbench.err
python3 tools/graph_backbone.py > $out/graph_backbone.json 2> $out/graph_backbone.err
for tk in 1 2; do python3 tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 1.5 3 8 --grid 1 --iters 20 --tile-kernel $tk; done > $out/opbench_tile_kernels_ab.jsonl 2> $out/opbench_ab.err
{ for s in 0.01 3 8; do EDGES="" bash tools/kstats_t3.sh $s 0; done; EDGES="" bash tools/kstats_t3.sh 0.01 1 2 4 6 7 8; } > $out/tile3_ablations.txt 2>&1
SNIPPER_DENSE_TABLE=$out/dense.json python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-locality-sweep > /dev/null 2> $out/dense.err; python3 tools/dense_roofline.py $out/dense.json > $out/backbone_roofline.csv
python3 tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 3 8 --grid 1 --iters 10 > $out/opbench_locality_bf16_value.jsonl 2> $out/opbench.err
python3 tools/opbench.py --N 8 --cases enc_local --dtypes float32 --skip-torch --rows-bf16 1 --sigma 0.01 3 8 --grid 1 --iters 10 > $out/opbench_locality_f32_value.jsonl 2>> $out/opbench.err
python3 tools/wresbench.py > $out/wresbench.jsonl 2> $out/wresbench.err
python3 tools/wgradbench.py > $out/wgradbench.jsonl 2> $out/wgradbench.err
bash tools/bench_kstats.sh $1 > $out/kstats.txt 2>&1
cp gpurun_out/kstats_$1.csv $out/kernel_stats.csv
python3 bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline --no-locality-sweep > $out/bench_fp32.json 2> $out/bench_fp32.err
SNIPPER_FORCE_DDP=1 SNIPPER_SYNC_FORCE=1 python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-locality-sweep > $out/bench_forced_ddp.json 2> $out/bench_forced_ddp.err
python3 bench.py --steps 20 --warmup 5 --pin-cores 0 --gc-every 0 --no-cpu-baseline --no-locality-sweep > $out/bench_host_untuned.json 2> $out/bench_host_untuned.err
python3 tools/convbench.py > $out/conv_ring_default.jsonl 2> $out/convbench.err
SNIPPER_CONV_RING=0 python3 tools/convbench.py > $out/conv_ring_off.jsonl 2>> $out/convbench.err
SNIPPER_CONV_RING=2 python3 tools/convbench.py > $out/conv_ring_all.jsonl 2>> $out/convbench.err
python3 bench.py > $out/bench_bf16_default.json 2> $out/bench_bf16_default.err
tail -c 900 $out/bench_bf16_default.json; echo; cat $out/copybench.json; head -5 $out/pmc_mfma_busy.csv; head -12 $out/pmc_lds_tcc.csv
