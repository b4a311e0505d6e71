mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_model_gpu.py -x -q -k "attn or agent or acting or model" 2>&1 | tail -5 > gpurun_out/t_m.log
timeout 300 python tools/policy_step_probe.py 40 2>&1 | tail -1 > gpurun_out/policy_m.txt
timeout 600 python tools/acting_probe.py > gpurun_out/acting_m.json 2>/dev/null
cat gpurun_out/t_m.log gpurun_out/policy_m.txt gpurun_out/acting_m.json
