#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_d; mkdir -p $O
# duo kernel: counters of the best setting found (profiles/r04/experiments/duo_kernel.txt)
export SRT_KERNEL_VARIANT=2 SRT_DUO_W_SWAP=16 SRT_DUO_W_BLOCKED=70 SRT_DUO_FILL_D=20 SRT_DUO_FILL_E=12 SRT_DUO_FILL_G=4
KERNEL_FILTER=render_kernel_duo bash tools/pmc_passes.sh $O/pmc_duo --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0 > $O/pmc_duo.log 2>&1
unset SRT_KERNEL_VARIANT SRT_DUO_W_SWAP SRT_DUO_W_BLOCKED SRT_DUO_FILL_D SRT_DUO_FILL_E SRT_DUO_FILL_G
cat $O/pmc_duo/summary.txt
# first-row assignment on / off, 1 GPU and W = 8, same box
for L in "" "$PWD/gpurun_exp_noassign.so"; do
  echo "== library: ${L:-default (first rows assigned)}" >> $O/assign_ab.txt
  SRT_LIB_PATH=$L python tools/world_emulation.py --worlds 1,8 --reps 4 2>&1 | grep "^world" >> $O/assign_ab.txt
  SRT_LIB_PATH=$L python tools/world_emulation.py --scene 100 --width 1280 --height 720 --spp 256 --worlds 1 --reps 4 2>&1 | grep "^world" >> $O/assign_ab.txt
done
cat $O/assign_ab.txt
