cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/solo
O=gpurun_out/solo
for ll in 1; do SRT_LIB_PATH=$PWD/gpurun_exp_diagshade.so SRT_DEBUG_LANE_LIMIT=$ll timeout -k 10 120 python tools/lone_tile.py --count 1 2>/dev/null; done | tee $O/lone_diagshade.txt
