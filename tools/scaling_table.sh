#!/bin/bash
# numbers behind DESIGN.md section 6 (run through gpurun): per-GPU time of rank 0's share, lone tile / pixel latencies
for W in 1 2 3 4 8; do
  for S in 0 default; do
    echo -n "world $W split_load $S: "; if [ $S = default ]; then unset SRT_SPLIT_LOAD; else export SRT_SPLIT_LOAD=$S; fi; timeout -k 10 120 python tools/diag.py --spp 1024 --world $W --rank 0 2>/dev/null | grep '"ms"'
  done
done
python tools/chain.py 2>/dev/null | grep "most expensive"
for p in 64 32 16 8 4 2 1; do SRT_TEST_KNOBS=1 SRT_DEBUG_LANE_LIMIT=$p timeout -k 10 100 python tools/lone_tile.py ${TILE_ARGS:-} 2>&1 | grep lane_limit; done
