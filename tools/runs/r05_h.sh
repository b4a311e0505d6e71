mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_fullsize_gpu.py -x -q -k "deterministic" 2>&1 | tail -5 > gpurun_out/t_h.log
timeout 600 python tools/northstar_probe.py > gpurun_out/ns_h.json 2> gpurun_out/ns_h.err
cat gpurun_out/t_h.log gpurun_out/ns_h.json; tail -n 3 gpurun_out/ns_h.err
