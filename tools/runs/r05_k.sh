mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preproc_gpu.py tests/test_kernels_gpu.py -x -q -k "preproc or vit or siglip or attn or mid_m" 2>&1 | tail -4 > gpurun_out/t_k.log
timeout 300 python tools/vit_probe.py > gpurun_out/vit_k.txt 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/v_kt; rocprofv3 --kernel-trace --stats -d /tmp/v_kt -o kt -- python $GRAFT_REPO_ROOT/tools/vit_probe.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summarize.py stats "$(find /tmp/v_kt -name '*.db' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/vit_k_stats.txt "vit probe" > /dev/null
cd $GRAFT_REPO_ROOT; cat gpurun_out/t_k.log gpurun_out/vit_k.txt; head -9 gpurun_out/vit_k_stats.txt | cut -c1-130
