#!/bin/bash
# whole-frame parity of the headline frame against the CPU oracle on the box's host cores (opt-in test of the GPU suite), on the
# SAH builder's tree and on the throughput-tuned tree.  usage (through gpurun): tools/_parity.sh <outdir>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r05_parity}; mkdir -p $O
for T in 0 1; do
  SRT_LONG=1 SRT_LONG_CFG=3 SRT_LONG_TUNED=$T SRT_LONG_THREADS=16 timeout -k 10 560 python -m pytest "tests/test_gpu_parity.py::test_full_frame_full_spp_bit_exact" -m gpu -x -q -s > $O/full_frame_cfg3_tuned$T.txt 2>&1; echo "cfg3 tuned=$T rc=$?"; tail -3 $O/full_frame_cfg3_tuned$T.txt
done
