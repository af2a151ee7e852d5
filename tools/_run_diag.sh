cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SRT_LIB_PATH=$PWD/gpurun_exp_diagshade.so timeout -k 10 300 python tools/diag.py --spp 256 > gpurun_out/diagshade.json 2>gpurun_out/diagshade.err
tail -40 gpurun_out/diagshade.json
