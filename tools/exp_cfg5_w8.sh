#!/bin/bash
# Where one rank of an 8-rank cfg 5 job (3840x2160 x 4096 spp) spends its GPU time next to the 1-GPU run: instrumented wave
# life times (steady / drain / idle-slot shares, lane utilisation per step kind) and the production kernel with the scheduling
# knobs switched off one at a time.  Usage (GPU box): bash tools/exp_cfg5_w8.sh <outdir> [spp]
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r04_cfg5w8}; SPP=${2:-4096}; mkdir -p $O
A="--scene 101 --width 3840 --height 2160 --spp $SPP"
echo "== instrumented, rank 0 of 8" > $O/tail.txt
timeout -k 10 600 python tools/wave_tail.py $A --world 8 --rank 0 --reps 1 >> $O/tail.txt 2>&1
echo "== instrumented, rank 5 of 8" >> $O/tail.txt
timeout -k 10 600 python tools/wave_tail.py $A --world 8 --rank 5 --reps 1 >> $O/tail.txt 2>&1
echo "== instrumented, 1 GPU, 512 spp" >> $O/tail.txt
timeout -k 10 600 python tools/wave_tail.py --scene 101 --width 3840 --height 2160 --spp 512 --reps 1 >> $O/tail.txt 2>&1
for K in "" "SRT_SPLIT_LOAD=0" "SRT_PROBE_SPP=0" "SRT_PROBE_SPP=8"; do
  echo "== production kernel, ranks 0,1 of 8, knobs: ${K:-default}" >> $O/knobs.txt
  env $K timeout -k 10 900 python tools/world_emulation.py $A --worlds 8 --ranks 0,1 >> $O/knobs.txt 2>&1
done
cat $O/tail.txt $O/knobs.txt
