#!/bin/bash
# round 5: the RCCL code path executed on one rank (SVLA_FORCE_DIST=1), bench.py timed the way the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -x -q -k "through_rccl" 2>&1 | tail -15 > gpurun_out/t_rccl1.log
cat gpurun_out/t_rccl1.log
# the headline workload with its one rank on the "nccl" backend: 3 asynchronous per-tower all-reduces per optimiser step through RCCL
SVLA_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r05_bench_rccl1.json 2> gpurun_out/bench_rccl1.err
tail -c 1500 gpurun_out/r05_bench_rccl1.json; tail -3 gpurun_out/bench_rccl1.err
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/bench_driver_style.err
E2=$(date +%s)
echo "driver-style bench wall: $((E2-S)) s" | tee gpurun_out/bench_driver_style_wall.txt
tail -3 gpurun_out/bench_driver_style.err; head -c 600 gpurun_out/r05_bench_driver_style.json
