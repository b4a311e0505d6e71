#!/bin/bash
# Run ON the GPU box: FETCH_SIZE of the TN weight-gradient kernel per shape (tools/pmc_tn.py N K; M = 8192*181) -> gpurun_out/<tag>_pmc_tn.txt
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/${TAG}_pmc_tn.txt
for nk in "512 512" "1536 512" "2048 512" "512 2048" "1024 512"; do
  rm -rf /tmp/ptn
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/ptn -o x -- python $R/tools/pmc_tn.py $nk > /dev/null 2> /tmp/ptn.err
  echo "N K = $nk" >> $R/gpurun_out/${TAG}_pmc_tn.txt
  python $R/tools/pmc_dump.py gemm_tn $(find /tmp/ptn -name "*.db") >> $R/gpurun_out/${TAG}_pmc_tn.txt 2>&1
done
