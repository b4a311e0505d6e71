#!/bin/bash
# round 5, final cycle on the final sources (kernel sources + SVLA_FORCE_DIST): IL throughput of three presets, full GPU suite with durations, stamped profiles, complete bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_il_throughput.txt
for v in small_3 siglip_base_3 siglip_base_6_6; do
  echo "== $v (python -m safevla_amd.train_il --model_version $v --per_gpu_batch 16 --sliding_window 50 --max_samples 128)" >> gpurun_out/r05_il_throughput.txt
  timeout 600 python -m safevla_amd.train_il --model_version $v --per_gpu_batch 16 --sliding_window 50 --max_samples 128 --output_dir /tmp/il_$v 2>/dev/null | tail -1 >> gpurun_out/r05_il_throughput.txt
done
cat gpurun_out/r05_il_throughput.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=45 2>&1 | tail -60 > gpurun_out/t_full6.log
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
for f in r05_kernel_stats.txt r05_pmc_hbm_traffic.json r05_pmc_hbm_traffic_by_shape.json r05_pmc_preproc_traffic.json r05_vit_kernel_stats.txt; do cp gpurun_out/$f profiles/$f; done
timeout 1500 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/bench_w.err
cat gpurun_out/t_full6.log; tail -n 8 gpurun_out/r05_profile_round.log; tail -n 2 gpurun_out/bench_w.err; tail -c 400 gpurun_out/r05_bench.json
