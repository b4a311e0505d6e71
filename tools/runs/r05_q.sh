#!/bin/bash
# round 5: the complete bench line on the final kernel sources (stamped profiles in profiles/ match)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/bench_q.err
tail -n 2 gpurun_out/bench_q.err; tail -c 600 gpurun_out/r05_bench.json
