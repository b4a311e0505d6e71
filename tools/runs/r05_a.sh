set -x
timeout 900 python -m pytest tests/test_c1_config_gpu.py -x -q -s 2>&1 | tail -15 > gpurun_out/t_c1.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "assembly" 2>&1 | tail -8 > gpurun_out/t_asm.log
SVLA_GEMM_LOG=/tmp/ga.log SVLA_GEMM_LOG_ALL=1 timeout 600 python tools/acting_probe.py > gpurun_out/acting0.json 2>/tmp/a.err; python tools/gemm_shapes.py /tmp/ga.log > gpurun_out/acting_shapes.txt
SVLA_GEMM_LOG=/tmp/gn.log SVLA_GEMM_LOG_ALL=1 timeout 600 python tools/northstar_probe.py > gpurun_out/ns0.json 2>/tmp/n.err; python tools/gemm_shapes.py /tmp/gn.log > gpurun_out/ns_shapes.txt
SVLA_GEMM_LOG=/tmp/gv.log SVLA_GEMM_LOG_ALL=1 timeout 600 python tools/vit_probe.py > gpurun_out/vit0.txt 2>/tmp/v.err; python tools/gemm_shapes.py /tmp/gv.log > gpurun_out/vit_shapes.txt
tail -3 /tmp/a.err /tmp/n.err /tmp/v.err
