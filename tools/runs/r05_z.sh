#!/bin/bash
# round 5: fuzz tests inside the suite + randomised engine configurations (tools/fuzz_engine.py) after the T = 1 fix
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fuzz_gpu.py tests/test_engine_gpu.py tests/test_agent_gpu.py tests/test_model_gpu.py tests/test_round2_gpu.py -m gpu -x -q 2>&1 | tail -5
: > gpurun_out/r05_fuzz_engine.txt
for s in 0 1 2; do timeout 1200 python tools/fuzz_engine.py --seed $s --cases 40 >> gpurun_out/r05_fuzz_engine.txt 2> gpurun_out/fuzz_engine.err; echo "seed $s rc $?"; done
grep -E "FAIL|failing" gpurun_out/r05_fuzz_engine.txt; tail -3 gpurun_out/fuzz_engine.err
