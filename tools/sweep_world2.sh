#!/bin/bash
for W in 8 4 3 2 1; do
  echo -n "world $W: "; timeout -k 10 120 python tools/diag.py --spp 1024 --world $W --rank 0 2>/dev/null | grep '"ms"'
done
