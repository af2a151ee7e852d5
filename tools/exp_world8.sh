#!/bin/bash
# W = 8 strong-scaling emulation for the default library and every gpurun_exp_*.so
cd "$GRAFT_REPO_ROOT"
echo default; python tools/world_emulation.py --worlds 8 --reps 2 2>&1 | grep "^world"
for f in gpurun_exp_*.so; do [ -f "$f" ] && { echo $f; SRT_LIB_PATH=$PWD/$f python tools/world_emulation.py --worlds 8 --reps 2 2>&1 | grep "^world"; }; done
