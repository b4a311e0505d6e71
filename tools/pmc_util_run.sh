#!/bin/bash
# Run ON the GPU box: two PMC passes over the default bench workload -> gpurun_out/<tag>_pmc_utilisation.json (tools/pmc_util_summary.py)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD0=${PMC_CMD:-"python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary"}      # PMC_CMD: another workload (e.g. tools/vit_probe.py)
rm -rf /tmp/pu1 /tmp/pu2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pu1 -o a -- $CMD0 > /dev/null 2> /tmp/pu1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d /tmp/pu2 -o b -- $CMD0 > /dev/null 2> /tmp/pu2.err
python $R/tools/pmc_util_summary.py $R/gpurun_out/${TAG}_pmc_utilisation.json $(find /tmp/pu1 /tmp/pu2 -name "*.db") > $R/gpurun_out/${TAG}_pmc_utilisation.txt 2>&1
