#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for P in 0 50 100 200; do
  echo "== SRT_ORDER_MAX_PCT=$P" >> $O/order_max.txt
  SRT_ORDER_MAX_PCT=$P python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 8 --ranks 0 2>&1 | grep "^world" >> $O/order_max.txt
  SRT_ORDER_MAX_PCT=$P python tools/world_emulation.py --worlds 1,8 --reps 3 2>&1 | grep "^world" >> $O/order_max.txt
done
cat $O/order_max.txt
echo "== instrumented tail, rank 0 of 8, SRT_ORDER_MAX_PCT=100" > $O/tail_order100.txt
SRT_ORDER_MAX_PCT=100 timeout -k 10 600 python tools/wave_tail.py --scene 101 --width 3840 --height 2160 --spp 4096 --world 8 --rank 0 --reps 1 2>&1 | grep -v amdgpu.ids | head -8 >> $O/tail_order100.txt
cat $O/tail_order100.txt
