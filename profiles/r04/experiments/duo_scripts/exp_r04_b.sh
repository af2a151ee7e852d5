#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_b; mkdir -p $O
( SRT_LIB_PATH=$PWD/gpurun_exp_duostats.so python tools/duo_stats.py; SRT_DUO_W_SWAP=64 SRT_DUO_W_BLOCKED=35 SRT_LIB_PATH=$PWD/gpurun_exp_duostats.so python tools/duo_stats.py ) 2>&1 | grep -v amdgpu.ids > $O/duo_stats.txt
timeout -k 10 600 python tools/duo_sweep.py --grid "W_SWAP=64,128,256,512;W_BLOCKED=18,35,70,140;FILL_D=24,32,48,64;FILL_E=12,16,24,32;FILL_G=8,64" 2>&1 | grep -v amdgpu.ids > $O/duo_sweep2.txt
tail -9 $O/duo_sweep2.txt
A="--scene 101 --width 3840 --height 2160 --spp 4096"
( echo "== default library"; python tools/world_emulation.py $A --worlds 8 --ranks 0; echo "== wave priorities compiled into the L2 variant"; SRT_LIB_PATH=$PWD/gpurun_exp_priol2.so python tools/world_emulation.py $A --worlds 8 --ranks 0,3 ) 2>&1 | grep -v amdgpu.ids > $O/cfg5_prio_l2.txt
cat $O/cfg5_prio_l2.txt
python tools/world_emulation.py --worlds 1,2,4,8 --reps 5 2>&1 | grep -v amdgpu.ids > $O/cfg3_world_reps.txt
cat $O/cfg3_world_reps.txt | grep "^world"
