mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_preproc_gpu.py -x -q -k "mid_m or patchify or preproc or vit or normalize or siglip" 2>&1 | tail -8 > gpurun_out/t_f.log
AB_EXTRA=1 timeout 600 python tools/ab_midm.py 2>&1 | grep -v SWEEP | tail -6 > gpurun_out/ab_f.txt
timeout 300 python tools/vit_probe.py > gpurun_out/vit_f.txt 2>&1
timeout 600 python tools/acting_probe.py > gpurun_out/acting_f.json 2>/dev/null
cat gpurun_out/t_f.log gpurun_out/ab_f.txt gpurun_out/vit_f.txt gpurun_out/acting_f.json
