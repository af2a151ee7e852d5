#!/bin/bash
echo -n "thresholds: "; timeout -k 10 120 python tools/diag.py --spp 1024 2>/dev/null | grep '"ms"'
for S in ${SH:-30 45 64 90}; do
  for F in ${FR:-64 100 150}; do
    echo -n "score shade $S fringe $F: "; SRT_SCORE_SHADE=$S SRT_SCORE_FRINGE=$F timeout -k 10 120 python tools/diag.py --spp 1024 2>/dev/null | grep '"ms"'
  done
done
