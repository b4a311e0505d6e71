#!/bin/bash
# round 5: randomised differential test of the kernel entry points (tools/fuzz_kernels.py), several seeds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_fuzz.txt
for s in 1 2 3 4 5 6; do
  timeout 900 python tools/fuzz_kernels.py --seed $s --cases 1500 >> gpurun_out/r05_fuzz.txt 2> gpurun_out/fuzz.err; echo "seed $s rc $?" | tee -a gpurun_out/r05_fuzz.txt
done
grep -E "FAIL|failure line|rc " gpurun_out/r05_fuzz.txt | head -80; tail -3 gpurun_out/fuzz.err
