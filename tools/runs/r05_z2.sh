#!/bin/bash
# round 5: engine fuzz with the deployment switches (fp8 attention, per-row T5 dropout, deterministic accumulation) at random
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_fuzz_engine.txt
for s in 4 5; do timeout 1200 python tools/fuzz_engine.py --seed $s --cases 40 >> gpurun_out/r05_fuzz_engine.txt 2> gpurun_out/fuzz_engine.err; echo "seed $s rc $?"; done
grep -E "FAIL|failing" gpurun_out/r05_fuzz_engine.txt; tail -3 gpurun_out/fuzz_engine.err; grep -c "fp8=1" gpurun_out/r05_fuzz_engine.txt; grep -c "deterministic=1" gpurun_out/r05_fuzz_engine.txt
