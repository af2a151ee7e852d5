#!/bin/bash
# Round 5's measurement round, in parts (one gpurun call each, <= 20 minutes).  usage: tools/r05_round.sh A|B|C|D [tag]
#   A  GPU suite, the default bench line (the driver's command shape), the headline part under rocprofv3 --kernel-trace --stats,
#      single-process N = 2 rehearsal over the test transport
#   B  SQ / TCC counter passes of cfg 3 on both trees + the second spp point of the fabric-traffic split
#   C  the same for cfg 4 (PRISM) and cfg 5's scene
#   D  strong-scaling emulation (cfg 3, cfg 5 at W = 8), instrumented phase split of cfg 5's scene at 128 spp
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${1:-A}; TAG=${2:-v29}; O=gpurun_out/r05_$TAG; mkdir -p $O
X="--steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0 --no-builders-tree"
L=$O/lane_ops_per_ray.json
case $PART in
A)
  timeout -k 10 600 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
  timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; cp -r gpurun_out/prof_$TAG $O/prof; tail -4 $O/profile_round.log
  SRT_RCCL_LIB=$PWD/tests/cpp/_build/libmock_rccl.so SRT_COMM_TEST_SAME_DEVICE=1 SRT_TEST_KNOBS=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --cfg5-spp 64 > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err; echo "bench --gpus 2 (rehearsal, one GPU, test transport) rc=$?"
  ;;
B)
  bash tools/pmc_passes.sh $O/pmc_cfg3 > $O/pmc_cfg3.log 2>&1
  PMC_ONLY=traffic bash tools/pmc_passes.sh $O/pmc_cfg3_256 --spp 256 $X > $O/pmc_cfg3_256.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg3 "$TAG" 100 $L --traffic $O/pmc_cfg3_256 > $O/lane_ops_cfg3.txt 2>&1; head -16 $O/lane_ops_cfg3.txt
  bash tools/pmc_passes.sh $O/pmc_cfg3b --spp 64 --no-profile-order $X > $O/pmc_cfg3b.log 2>&1
  PMC_ONLY=traffic bash tools/pmc_passes.sh $O/pmc_cfg3b_256 --spp 256 --no-profile-order $X > $O/pmc_cfg3b_256.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg3b "$TAG" 100 $L --traffic $O/pmc_cfg3b_256 --key scene_100_builders_tree > $O/lane_ops_cfg3b.txt 2>&1; head -5 $O/lane_ops_cfg3b.txt
  ;;
C)
  bash tools/pmc_passes.sh $O/pmc_cfg5 --scene 101 --width 3840 --height 2160 --spp 32 $X > $O/pmc_cfg5.log 2>&1
  PMC_ONLY=traffic bash tools/pmc_passes.sh $O/pmc_cfg5_128 --scene 101 --width 3840 --height 2160 --spp 128 $X > $O/pmc_cfg5_128.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg5 "$TAG" 101 $L --traffic $O/pmc_cfg5_128 > $O/lane_ops_cfg5.txt 2>&1; head -16 $O/lane_ops_cfg5.txt
  bash tools/pmc_passes.sh $O/pmc_cfg4 --scene 1 --bvh 0 --spp 128 $X > $O/pmc_cfg4.log 2>&1
  PMC_ONLY=traffic bash tools/pmc_passes.sh $O/pmc_cfg4_512 --scene 1 --bvh 0 --spp 512 $X > $O/pmc_cfg4_512.log 2>&1
  python tools/pmc_to_lane_ops.py $O/pmc_cfg4 "$TAG" 1 $L --traffic $O/pmc_cfg4_512 > $O/lane_ops_cfg4.txt 2>&1; head -5 $O/lane_ops_cfg4.txt
  bash tools/pmc_mem.sh $O/pmc_mem_cfg5 --scene 101 --width 3840 --height 2160 --spp 32 $X > $O/pmc_mem_cfg5.log 2>&1; tail -30 $O/pmc_mem_cfg5.log
  ;;
D)
  timeout -k 10 300 python tools/diag.py --scene 101 --width 3840 --height 2160 --spp 128 --reps 1 > $O/diag_cfg5_128spp.json 2>&1
  timeout -k 10 300 python tools/diag.py --spp 64 > $O/diag_cfg3_64spp.json 2>&1
  timeout -k 10 500 python tools/world_emulation.py --worlds 1,2,4,8 --reps 3 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg3.txt; grep "^world" $O/world_emulation_cfg3.txt
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse-gloo --spp 64 --steps 2 --warmup 1 --cfg5-spp 16 --cfg5-size 640x360 > $O/bench_torchrun2_gloo_rehearsal.json 2> $O/bench_torchrun2_gloo_rehearsal.err; echo "torchrun rehearsal rc=$?"
  ;;
E)
  # cfg 5 as specified on one rank of eight (all eight, one after the other) and on one GPU: the strong-scaling emulation of the largest configuration
  timeout -k 10 1100 python tools/world_emulation.py --scene 101 --width 3840 --height 2160 --spp 4096 --worlds 1,8 --reps 1 2>&1 | grep -v amdgpu.ids > $O/world_emulation_cfg5_4096spp.txt; grep "^world" $O/world_emulation_cfg5_4096spp.txt
  ;;
F)
  # whole frame of cfg 5's scene (3840x2160 x 8 spp: 18 768 blocks) and of cfg 2 against the CPU oracle
  SRT_LONG=1 SRT_LONG_CFG=5 SRT_LONG_SPP=8 SRT_LONG_THREADS=16 timeout -k 10 560 python -m pytest "tests/test_gpu_parity.py::test_full_frame_full_spp_bit_exact" -m gpu -x -q -s > $O/full_frame_cfg5_8spp.txt 2>&1; echo "cfg5 rc=$?"; tail -3 $O/full_frame_cfg5_8spp.txt
  SRT_LONG=1 SRT_LONG_CFG=2 SRT_LONG_THREADS=16 timeout -k 10 400 python -m pytest "tests/test_gpu_parity.py::test_full_frame_full_spp_bit_exact" -m gpu -x -q -s > $O/full_frame_cfg2.txt 2>&1; echo "cfg2 rc=$?"; tail -3 $O/full_frame_cfg2.txt
  ;;
esac
rm -rf $O/pmc_*/pass*/
