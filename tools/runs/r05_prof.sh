mkdir -p gpurun_out
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
tail -n 30 gpurun_out/r05_profile_round.log
