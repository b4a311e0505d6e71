#!/bin/bash
# round 5, last call: the driver's own bench command on HEAD + two more seeds of the acting / kernel campaigns
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/bench_driver_style.err
echo "driver-style bench wall: $(( $(date +%s) - S )) s"
timeout 300 python tools/fuzz_acting.py --seed 7 --cases 9 2>/dev/null | tail -3
timeout 300 python tools/fuzz_kernels.py --seed 21 --cases 2000 2>/dev/null | tail -2
tail -2 gpurun_out/bench_driver_style.err; head -c 300 gpurun_out/r05_bench_driver_style.json
