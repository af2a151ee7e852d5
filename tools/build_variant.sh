#!/bin/bash
# builds tools/_variants/<name>.so (git-ignored; travels to the GPU box with gpurun) from the current sources with extra compiler
# flags for srt_kernels.hip, and prints the register budget of the production render kernels.  Load it with SRT_LIB_PATH.
# usage: tools/build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../cuda-spectral-ray-tracer_amd/csrc"
name=$1; shift
mkdir -p ../../tools/_variants
F="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
/opt/rocm/bin/hipcc $F "$@" -c srt_kernels.hip -o /tmp/k_$name.o 2>/tmp/k_$name.err || { grep -i -A4 "error" /tmp/k_$name.err | head -20; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_variants/$name.so /tmp/k_$name.o _build/srt_capi.o _build/srt_host.o _build/srt_calib.o _build/srt_comm.o -ldl
for k in ILi0ELb1ELb1 ILi0ELb0ELb0 ILi0ELb1ELb0; do
  grep -A12 "Function Name: .*render_kernel$k" /tmp/k_$name.err | grep "VGPRs:\|VGPRs Spill\|SGPRs Spill\|Scratch" | tr -s ' ' | tr '\n' ' ' | sed "s/remark: srt_kernels.hip:[0-9]*:0://g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/^/$k: /"; echo
done
echo "built tools/_variants/$name.so"
