#!/bin/bash
# round 5, re-entry: full GPU suite + smoke on HEAD (96-wide IL presets, IL agent), attention A/B timing baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t_full5.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke5.log 2>&1
timeout 200 python tools/ab_attn.py > gpurun_out/ab_attn5.log 2>&1
cat gpurun_out/t_full5.log; tail -2 gpurun_out/smoke5.log; tail -20 gpurun_out/ab_attn5.log
