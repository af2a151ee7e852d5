cd "$GRAFT_REPO_ROOT"
bash tools/r02_round.sh v22 > gpurun_out/r02_v22.log 2>&1; tail -45 gpurun_out/r02_v22.log | head -5
bash tools/profile_round.sh v22 > gpurun_out/prof_v22.log 2>&1; tail -14 gpurun_out/prof_v22.log
