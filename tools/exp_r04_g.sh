#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_g; mkdir -p $O
for L in "" "$PWD/gpurun_exp_perm1.so" "$PWD/gpurun_exp_perm2.so" "$PWD/gpurun_exp_noassign.so"; do
  echo "== library: ${L:-default (bands in wave order)}" >> $O/assign_perm.txt
  SRT_LIB_PATH=$L python tools/world_emulation.py --worlds 1,4,8 --reps 4 2>&1 | grep "^world" >> $O/assign_perm.txt
  SRT_LIB_PATH=$L python tools/world_emulation.py --width 1280 --height 720 --spp 256 --worlds 1 --reps 4 2>&1 | grep "^world" >> $O/assign_perm.txt
done
cat $O/assign_perm.txt
echo "== cfg 5 at W = 8, rank 0 (automatic queue order)" > $O/cfg5_auto.txt
python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 8 --ranks 0 2>&1 | grep "^world" >> $O/cfg5_auto.txt
cat $O/cfg5_auto.txt
for SH in 55 70 85 100; do for FR in 240 280 320; do
  echo -n "score_shade $SH score_fringe $FR: " >> $O/scores.txt
  SRT_SCORE_SHADE=$SH SRT_SCORE_FRINGE=$FR python tools/world_emulation.py --worlds 1 --reps 2 2>&1 | grep "^world" >> $O/scores.txt
done; done
cat $O/scores.txt
