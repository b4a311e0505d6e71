mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/t_full2.log
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
timeout 1200 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/bench_j.err
cat gpurun_out/t_full2.log; tail -n 12 gpurun_out/r05_profile_round.log; tail -n 3 gpurun_out/bench_j.err
