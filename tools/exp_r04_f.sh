#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_f; mkdir -p $O
A="--scene 101 --width 3840 --height 2160 --spp 4096"
for K in "SRT_ORDER_MAX_PCT=100" "SRT_ORDER_MAX_PCT=100 SRT_SPLIT_BY_KEY=1" "SRT_ORDER_MAX_PCT=100 SRT_SPLIT_BY_KEY=1 SRT_SPLIT_LOAD=150" "SRT_ORDER_MAX_PCT=100 SRT_SPLIT_BY_KEY=1 SRT_SPLIT_LOAD=100" "SRT_ORDER_MAX_PCT=100 SRT_SPLIT_BY_KEY=1 SRT_SPLIT_LOAD=70" "SRT_ORDER_MAX_PCT=150 SRT_SPLIT_BY_KEY=1 SRT_SPLIT_LOAD=100"; do
  echo "== $K" >> $O/split_key.txt
  env $K python tools/world_emulation.py $A --worlds 8 --ranks 0 2>&1 | grep "^world" >> $O/split_key.txt
done
for K in "SRT_ORDER_MAX_PCT=0" "SRT_ORDER_MAX_PCT=100" "SRT_ORDER_MAX_PCT=100 SRT_SPLIT_BY_KEY=1"; do
  echo "== $K  (cfg 2 on 1 GPU, cfg 3 at W = 2, 4, 8)" >> $O/split_key.txt
  env $K python tools/world_emulation.py --width 1280 --height 720 --spp 256 --worlds 1 --reps 3 2>&1 | grep "^world" >> $O/split_key.txt
  env $K python tools/world_emulation.py --worlds 2,4,8 --reps 3 2>&1 | grep "^world" >> $O/split_key.txt
done
cat $O/split_key.txt
