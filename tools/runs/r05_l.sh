mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o x -- python $R/tools/policy_step_probe.py 40 > $R/gpurun_out/policy_probe.log 2>&1
python $R/tools/prof_summarize.py stats "$(find /tmp/pp -name '*.db' | head -1)" $R/gpurun_out/r05_policy_step_kernel_stats.txt "KV-cached 3-tower acting step, 64 envs, 44 steps + rollout fill (tools/policy_step_probe.py)" > /dev/null
tail -n 2 $R/gpurun_out/policy_probe.log; head -40 $R/gpurun_out/r05_policy_step_kernel_stats.txt | cut -c1-140
