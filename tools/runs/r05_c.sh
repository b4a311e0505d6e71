mkdir -p gpurun_out
SVLA_NT_AS_MIN_PANELS=1 AB_SWEEP=1 timeout 1200 python tools/ab_midm.py > gpurun_out/ab_midm2.txt 2>&1
