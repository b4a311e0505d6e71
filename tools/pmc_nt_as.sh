#!/bin/bash
# PMC passes over the assembly NT kernel (one counter group per run; no tracing domains beside --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-2048}
OUT=$R/gpurun_out/pmc_nt_as_$N
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" \
           "TA_BUSY_avg TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- python $R/tools/run_nt_as_once.py $N > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in acc.items():
        if "nt_as" in k or "gemm" in k:
            print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
