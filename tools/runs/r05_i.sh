mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preproc_gpu.py tests/test_agent_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/t_i.log
timeout 300 python tools/vit_probe.py > gpurun_out/vit_i.txt 2>&1
timeout 600 python tools/acting_probe.py > gpurun_out/acting_i.json 2>/dev/null
cat gpurun_out/t_i.log gpurun_out/vit_i.txt gpurun_out/acting_i.json
