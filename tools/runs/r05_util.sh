mkdir -p gpurun_out
bash tools/pmc_util_run.sh r05
PMC_CMD="python $GRAFT_REPO_ROOT/tools/vit_probe.py" bash tools/pmc_util_run.sh r05_vit
PMC_CMD="python $GRAFT_REPO_ROOT/tools/northstar_probe.py" bash tools/pmc_util_run.sh r05_northstar
cat gpurun_out/r05_pmc_utilisation.txt | head -30; cat gpurun_out/r05_vit_pmc_utilisation.txt; cat gpurun_out/r05_northstar_pmc_utilisation.txt | head -20
