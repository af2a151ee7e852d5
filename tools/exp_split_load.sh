#!/bin/bash
# cfg 5 (3840x2160 x 4096 spp), rank 0 (and 1) of 8: the split policy's load factor.  At the default (200 %) nothing is split and the
# tiles of the glass object run for the whole launch while half of the wave slots idle (profiles/r04/cfg5_w8_tail.txt).
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r04_split}; mkdir -p $O; : > $O/split_load.txt
A="--scene 101 --width 3840 --height 2160 --spp ${2:-4096}"
for L in 200 150 120 100 80 60 40; do
  echo "== SRT_SPLIT_LOAD=$L" >> $O/split_load.txt
  SRT_SPLIT_LOAD=$L timeout -k 10 600 python tools/world_emulation.py $A --worlds 8 --ranks ${3:-0} 2>&1 | grep -v amdgpu.ids >> $O/split_load.txt
done
cat $O/split_load.txt
