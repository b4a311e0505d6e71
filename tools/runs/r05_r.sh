#!/bin/bash
# round 5: 768-wide IL presets -- width-parametric kernels and tower
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "small_linear or norm_fwd_bwd_plain or norm_adapter or rows_add or fusion_fill or decoder_embed" 2>&1 | tail -15 > gpurun_out/r05_r_k.txt
cat gpurun_out/r05_r_k.txt
timeout 2400 python -m pytest tests/test_il_gpu.py -x -q -m gpu -s 2>&1 | tail -30 > gpurun_out/r05_r_il.txt
cat gpurun_out/r05_r_il.txt
