#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python tools/world_emulation.py --worlds 1,2,4,8 --reps 5 2>&1 | grep -v amdgpu.ids > $O/cfg3_world_reps_assigned.txt
grep "^world" $O/cfg3_world_reps_assigned.txt
python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 8 --ranks 0,5 2>&1 | grep -v amdgpu.ids > $O/cfg5_w8_assigned.txt
grep "^world" $O/cfg5_w8_assigned.txt
timeout -k 10 600 python tools/duo_sweep.py --grid "W_SWAP=16,32,64;W_BLOCKED=35,70,100;FILL_D=20,24,28;FILL_E=4,8,12,16;FILL_G=2,4,8" 2>&1 | grep -v amdgpu.ids > $O/duo_sweep3.txt
tail -9 $O/duo_sweep3.txt
