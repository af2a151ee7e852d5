cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_m; mkdir -p $O
A="--scene 101 --width 3840 --height 2160 --spp 4096"
for P in 0 100; do for W in 4 2; do
  echo "== W=$W SRT_ORDER_MAX_PCT=$P" >> $O/order_w.txt
  SRT_ORDER_MAX_PCT=$P python tools/world_emulation.py $A --worlds $W --ranks 0 2>&1 | grep "^world" >> $O/order_w.txt
done; done
cat $O/order_w.txt
