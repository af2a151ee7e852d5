#!/bin/bash
# threshold sweep on the headline workload (run through gpurun); prints kernel ms per setting
for S in 24 32 40 48; do
  for F in 8 12 16 20 26; do
    echo -n "shade $S fringe $F: "; SRT_SHADE_THRESHOLD=$S SRT_FRINGE_THRESHOLD=$F timeout -k 10 120 python tools/diag.py --spp 1024 2>/dev/null | grep '"ms"'
  done
done
