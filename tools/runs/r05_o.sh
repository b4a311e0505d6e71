#!/bin/bash
# round 5: IL presets with the SigLIP text tower / wider image features
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_il_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r05_o_il.txt
cat gpurun_out/r05_o_il.txt
