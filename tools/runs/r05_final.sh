mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t_final.log
timeout 1500 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/bench_final.err
cat gpurun_out/t_final.log; tail -n 2 gpurun_out/bench_final.err
