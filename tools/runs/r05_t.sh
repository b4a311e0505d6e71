#!/bin/bash
# round 5: heads of 96 (base_6, siglip_base_3_6) through the fp32 attention kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fp32_mode_gpu.py -x -q -m gpu -k "attn_f32" 2>&1 | tail -8 > gpurun_out/r05_t_k.txt
cat gpurun_out/r05_t_k.txt
timeout 2400 python -m pytest tests/test_il_gpu.py -x -q -m gpu -s -k "wider or other_model" 2>&1 | tail -25 > gpurun_out/r05_t_il.txt
cat gpurun_out/r05_t_il.txt
