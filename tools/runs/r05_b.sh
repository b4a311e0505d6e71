mkdir -p gpurun_out
timeout 900 python tools/ab_midm.py > gpurun_out/ab_midm.txt 2>&1
timeout 600 python tools/acting_probe.py > gpurun_out/acting1.json 2>/tmp/a.err
timeout 600 python tools/northstar_probe.py > gpurun_out/ns1.json 2>/tmp/n.err
timeout 600 python tools/vit_probe.py > gpurun_out/vit1.txt 2>/tmp/v.err
tail -n 3 /tmp/a.err /tmp/n.err /tmp/v.err
