#!/bin/bash
# round 5: CU-partition probe (tools/cu_split_probe.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python tools/cu_split_probe.py > gpurun_out/r05_cu_split_probe.txt 2> gpurun_out/cu_split.err
echo "rc $?"; cat gpurun_out/r05_cu_split_probe.txt; tail -5 gpurun_out/cu_split.err
