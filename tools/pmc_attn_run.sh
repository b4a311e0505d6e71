#!/bin/bash
# Run ON the GPU box: PMC passes over tools/pmc_attn.py bwd (fusion-shape attention, R = 8192, S = 181): utilisation counters and
# HBM traffic of the attention kernels -> gpurun_out/<tag>_pmc_attn.txt
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pa$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pa$i -o x -- python $R/tools/pmc_attn.py bwd > /dev/null 2> /tmp/pa$i.err
done
python $R/tools/pmc_dump.py attn_ $(find /tmp/pa1 /tmp/pa2 /tmp/pa3 /tmp/pa4 /tmp/pa5 -name "*.db") > $R/gpurun_out/${TAG}_pmc_attn.txt 2>&1
