set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_t3; mkdir -p $O
echo "== weight sweep cfg3 (priorities on)"; bash tools/sweep_scores.sh "55 70 90" "220 280 360" --no-calibration
echo "== cfg 5 world emulation at the real workload (3840x2160 x 4096 spp)"
python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 1,8 > $O/world_emulation_cfg5_4096spp.txt 2>&1; grep "^world" $O/world_emulation_cfg5_4096spp.txt
echo "== cfg 3 world emulation"
python tools/world_emulation.py --worlds 1,2,4,8 > $O/world_emulation_cfg3.txt 2>&1; grep "^world" $O/world_emulation_cfg3.txt
