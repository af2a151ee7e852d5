#!/bin/bash
# one measurement round on the GPU box (through gpurun): GPU suite, the headline part of bench.py under rocprofv3 --kernel-trace
# --stats, HBM traffic counters, the default bench line, the single-process N = 2 rehearsal over the test transport, SQ counter
# passes of the three kernel variants / scenes bench.py prices (-> lane_ops_per_ray.json, tied to the kernel by the hash of its
# machine code), instrumented phase split, strong-scaling emulation of cfg 3 and cfg 5.  usage: tools/r04_round.sh <tag>
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-vX}; O=gpurun_out/r04_$TAG; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; cp -r gpurun_out/prof_$TAG $O/prof; tail -4 $O/profile_round.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
SRT_RCCL_LIB=$PWD/tests/cpp/_build/libmock_rccl.so SRT_COMM_TEST_SAME_DEVICE=1 SRT_TEST_KNOBS=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --cfg5-spp 64 > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err; echo "bench --gpus 2 (rehearsal, one GPU, test transport) rc=$?"
python tools/diag.py --spp 64 > $O/diag_cfg3_64spp.json 2>&1
bash tools/pmc_passes.sh $O/pmc_cfg3 > $O/pmc_cfg3.log 2>&1
python tools/pmc_to_lane_ops.py $O/pmc_cfg3 "$TAG" 100 $O/lane_ops_per_ray.json > $O/lane_ops_cfg3.txt 2>&1; head -14 $O/lane_ops_cfg3.txt
X="--steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0"
bash tools/pmc_passes.sh $O/pmc_cfg5 --scene 101 --width 3840 --height 2160 --spp 32 $X > $O/pmc_cfg5.log 2>&1
python tools/pmc_to_lane_ops.py $O/pmc_cfg5 "$TAG" 101 $O/lane_ops_per_ray.json > $O/lane_ops_cfg5.txt 2>&1
bash tools/pmc_passes.sh $O/pmc_cfg4 --scene 1 --bvh 0 --spp 128 $X > $O/pmc_cfg4.log 2>&1
python tools/pmc_to_lane_ops.py $O/pmc_cfg4 "$TAG" 1 $O/lane_ops_per_ray.json > $O/lane_ops_cfg4.txt 2>&1
python tools/world_emulation.py --worlds 1,2,4,8 --reps 3 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg3.txt; grep "^world" $O/world_emulation_cfg3.txt
python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 1,8 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg5_4096spp.txt; grep "^world" $O/world_emulation_cfg5_4096spp.txt
python tools/wave_tail.py --scene 101 --width 3840 --height 2160 --spp 4096 --world 8 --rank 0 --reps 1 2>&1 | grep -v amdgpu.ids > $O/cfg5_w8_tail_final.txt
rm -rf $O/pmc_*/pass*/
