#!/bin/bash
# ON the GPU box: kernel-trace statistics of the recorded acting step (tools/policy_step_probe.py), steady-state steps only.
#   bash tools/prof_policy_step.sh <tag> [env assignments, e.g. SVLA_GROUPED_TOWERS=0]
set -u
TAG=${1:-rXX}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_kt
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/ps_kt -o kt -- python $REPO/tools/policy_step_probe.py 44 > $OUT/${TAG}_policy_step.log 2> /tmp/ps.err
DB=$(find /tmp/ps_kt -name "*.db" | head -1)
python $REPO/tools/prof_summarize.py stats "$DB" $OUT/${TAG}_policy_step_kernel_stats.txt "KV-cached 3-tower acting step, 64 envs, 44 steps + rollout fill (tools/policy_step_probe.py; $TAG $*)" > /dev/null
python $REPO/tools/prof_summarize.py timeline "$DB" $OUT/${TAG}_policy_step_timeline.txt 130
tail -2 $OUT/${TAG}_policy_step.log
