#!/bin/bash
# per-GPU time of rank 0's share of a W-rank job for several split load factors (run through gpurun)
for W in ${WORLDS:-8 4}; do
  for S in ${LOADS:-0 150 220 300}; do
    echo -n "world $W split_load $S: "; SRT_SPLIT_LOAD=$S timeout -k 10 120 python tools/diag.py --spp 1024 --world $W --rank 0 2>/dev/null | grep '"ms"'
  done
done
