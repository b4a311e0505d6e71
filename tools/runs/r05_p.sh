#!/bin/bash
# round 5: kernel trace of the batch-256 probe (north_star secondary), then the full validation cycle (tests, stamped profiles, bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ns_kt && rocprofv3 --kernel-trace --stats -d /tmp/ns_kt -o kt -- python $GRAFT_REPO_ROOT/tools/northstar_probe.py > $GRAFT_REPO_ROOT/gpurun_out/ns_p.json 2> /tmp/ns.err
  python $GRAFT_REPO_ROOT/tools/prof_summarize.py stats "$(find /tmp/ns_kt -name '*.db' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r05_northstar_kernel_stats.txt "batch-256 probe (tools/northstar_probe.py: recording pass + 14 replays of the 3-tower fwd + bwd at 256 rows x 233 tokens, then the ViT leg; r05)" > /dev/null )
head -30 gpurun_out/r05_northstar_kernel_stats.txt
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/t_full3.log
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
timeout 1200 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/bench_p.err
cat gpurun_out/t_full3.log; tail -n 12 gpurun_out/r05_profile_round.log; tail -n 3 gpurun_out/bench_p.err
