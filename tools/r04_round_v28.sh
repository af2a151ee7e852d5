#!/bin/bash
# The measurement round after the tree tuning went in (same steps as tools/r04_round.sh, in two halves because one gpurun call is
# limited to 20 minutes).  usage: tools/r04_round_v28.sh A|B
set -u
cd "$GRAFT_REPO_ROOT"
TAG=v28; O=gpurun_out/r04_$TAG; mkdir -p $O
X="--steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0"
if [ "${1:-A}" = "A" ]; then
  python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
  bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; cp -r gpurun_out/prof_$TAG $O/prof; tail -4 $O/profile_round.log
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  SRT_RCCL_LIB=$PWD/tests/cpp/_build/libmock_rccl.so SRT_COMM_TEST_SAME_DEVICE=1 SRT_TEST_KNOBS=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --cfg5-spp 64 > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err; echo "bench --gpus 2 (rehearsal, one GPU, test transport) rc=$?"
  python tools/diag.py --spp 64 > $O/diag_cfg3_64spp.json 2>&1
  bash tools/pmc_passes.sh $O/pmc_cfg3 > $O/pmc_cfg3.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg3 "$TAG" 100 $O/lane_ops_per_ray.json > $O/lane_ops_cfg3.txt 2>&1; head -14 $O/lane_ops_cfg3.txt
else
  bash tools/pmc_passes.sh $O/pmc_cfg5 --scene 101 --width 3840 --height 2160 --spp 32 $X > $O/pmc_cfg5.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg5 "$TAG" 101 $O/lane_ops_per_ray.json > $O/lane_ops_cfg5.txt 2>&1; head -5 $O/lane_ops_cfg5.txt
  bash tools/pmc_passes.sh $O/pmc_cfg4 --scene 1 --bvh 0 --spp 128 $X > $O/pmc_cfg4.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg4 "$TAG" 1 $O/lane_ops_per_ray.json > $O/lane_ops_cfg4.txt 2>&1; head -5 $O/lane_ops_cfg4.txt
  python tools/world_emulation.py --worlds 1,2,4,8 --reps 3 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg3.txt; grep "^world" $O/world_emulation_cfg3.txt
  python tools/world_emulation.py --width 1280 --height 720 --spp 256 --worlds 1 --reps 5 2>&1 | grep -v amdgpu.ids > $O/cfg2_frames.txt; grep "^world" $O/cfg2_frames.txt
  SRT_TOOL_NO_TUNING=1 python tools/world_emulation.py --worlds 1 --reps 3 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg3_untuned_tree.txt; grep "^world" $O/world_emulation_cfg3_untuned_tree.txt
fi
rm -rf $O/pmc_*/pass*/
