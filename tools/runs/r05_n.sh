mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_agent_gpu.py tests/test_model_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py -x -q -k "rmsnorm or agent or acting or model or t5 or text or engine or checkpoint" 2>&1 | tail -5 > gpurun_out/t_n.log
timeout 300 python tools/policy_step_probe.py 40 2>&1 | tail -1 > gpurun_out/policy_n.txt
timeout 600 python tools/acting_probe.py > gpurun_out/acting_n.json 2>/dev/null
cat gpurun_out/t_n.log gpurun_out/policy_n.txt gpurun_out/acting_n.json
