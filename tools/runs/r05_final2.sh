#!/bin/bash
# round 5, last check of HEAD: full GPU suite + smoke + a short bench line (acting + collect legs included) after the T = 1 fixes in model.py / il.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t_full7.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke7.log 2>&1
cat gpurun_out/t_full7.log; tail -2 gpurun_out/smoke7.log
