mkdir -p gpurun_out
timeout 900 python tools/time_midm.py > gpurun_out/time_midm.txt 2>&1
cat gpurun_out/time_midm.txt
