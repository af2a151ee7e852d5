#!/bin/bash
# builds gpurun_exp_<name>.so in the repo root from the current sources with extra compiler flags for srt_kernels.hip
# usage: tools/build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../cuda-spectral-ray-tracer_amd/csrc"
name=$1; shift
F="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F "$@" -c srt_kernels.hip -o /tmp/k_$name.o 2>/tmp/k_$name.err || { grep -i -A4 "error" /tmp/k_$name.err | head -20; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_exp_$name.so /tmp/k_$name.o _build/srt_capi.o _build/srt_host.o _build/srt_calib.o _build/srt_comm.o -ldl
echo "built gpurun_exp_$name.so"
