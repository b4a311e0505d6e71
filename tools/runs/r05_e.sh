mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/t_full.log
timeout 900 python bench.py > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
tail -n 5 gpurun_out/bench_e.err
