#!/bin/bash
# usage: tools/copy_profiles.sh r05   -- copies what tools/collect_profiles.sh left under gpurun_out/<round>/ into profiles/<round>_*
r=$1; src=gpurun_out/$r; dst=profiles
cp $src/pmc_bench_step.csv            $dst/${r}_pmc_bench_step.csv
cp $src/kernel_stats.csv              $dst/${r}_bench_n1_bf16_kernel_stats.csv
cp $src/backbone_roofline.csv         $dst/${r}_backbone_roofline.csv
cp $src/bench_bf16_default.json       $dst/${r}_bench_n1_bf16_default_run.json
cp $src/bench_forced_ddp.json         $dst/${r}_bench_forced_ddp_1rank.json
cp $src/bench_host_untuned.json       $dst/${r}_bench_n1_bf16_host_untuned.json
cp $src/bench_fp32.json               $dst/${r}_bench_n1_fp32.json
cp $src/pmc_mfma_busy.csv             $dst/${r}_pmc_mfma_busy.csv
cp $src/pmc_lds_tcc.csv               $dst/${r}_pmc_lds_tcc.csv
cp $src/tile3_ablations.txt           $dst/${r}_tile3_ablations.txt
cp $src/copybench.json                $dst/${r}_hbm_copy_ceiling.json
cp $src/wgradbench.jsonl              $dst/${r}_wgradbench.jsonl
cp $src/wresbench.jsonl               $dst/${r}_wresbench.jsonl
cp $src/opbench_locality_bf16_value.jsonl $dst/${r}_opbench_locality_bf16_value.jsonl
cp $src/opbench_locality_f32_value.jsonl  $dst/${r}_opbench_locality_f32_value.jsonl
cp $src/opbench_tile_kernels_ab.jsonl $dst/${r}_opbench_tile_kernels_ab.jsonl
cp $src/graph_backbone.json           $dst/${r}_graph_backbone.json
cp $src/graphbench.jsonl              $dst/${r}_graphbench.jsonl
cat $src/conv_ring_default.jsonl $src/conv_ring_off.jsonl $src/conv_ring_all.jsonl > $dst/${r}_conv_ring_ab.jsonl
[ -f $src/kstats.txt ] && cp $src/kstats.txt $dst/${r}_bench_n1_bf16_kernel_summary.txt
ls -la $dst | grep ${r}_ | wc -l
