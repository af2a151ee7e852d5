#!/bin/bash
# one measurement round on the GPU box (through gpurun): default bench under rocprofv3 --kernel-trace --stats, HBM traffic counters,
# SQ counter passes of the headline scene (-> lane_ops_per_ray.json, tied to the kernel by the ISA hash), cfg 2 / cfg 5 counter
# passes, the other BASELINE configs as bench lines, instrumented phase splits.  usage: tools/r03_round.sh <tag>
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-vX}; O=gpurun_out/r03_$TAG; mkdir -p $O
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; cp -r gpurun_out/prof_$TAG $O/prof; tail -4 $O/profile_round.log
python tools/diag.py --spp 64 > $O/diag_cfg3_64spp.json 2>&1
bash tools/pmc_passes.sh $O/pmc_cfg3 > $O/pmc_cfg3.log 2>&1
python tools/pmc_to_lane_ops.py $O/pmc_cfg3 "$TAG" 100 $O/lane_ops_per_ray.json > $O/lane_ops_cfg3.txt 2>&1; head -12 $O/lane_ops_cfg3.txt
PASS_ARGS="--scene 100 --width 1280 --height 720 --spp 256 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration"
bash tools/pmc_passes.sh $O/pmc_cfg2 $PASS_ARGS > $O/pmc_cfg2.log 2>&1
PASS_ARGS="--scene 101 --width 3840 --height 2160 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration"
bash tools/pmc_passes.sh $O/pmc_cfg5 $PASS_ARGS > $O/pmc_cfg5.log 2>&1
python tools/pmc_to_lane_ops.py $O/pmc_cfg5 "$TAG" 101 $O/lane_ops_per_ray.json > $O/lane_ops_cfg5.txt 2>&1
PMC=0 bash tools/other_configs.sh $O/other > $O/other.log 2>&1; head -4 $O/other.log
python tools/diag.py --scene 100 --width 1280 --height 720 --spp 256 > $O/diag_cfg2.json 2>&1
rm -rf $O/pmc_*/pass*/ 
