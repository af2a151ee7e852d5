#!/usr/bin/env python3
"""Identity of the render kernel's machine code: sha256 of the production render kernels' bodies in the gfx950 ISA listing the build
keeps next to the object (csrc/_build/srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s, from -save-temps), comments and file / ident
directives removed.

The PMC passes (tools/pmc_to_lane_ops.py) store it next to the per-ray figures they derive; bench.py recomputes it for the
library it is timing and marks an imported figure whose hash differs as STALE.  Usage: python tools/kernel_id.py"""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
LIB = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "libsrt_hip.so")


RENDER = re.compile(r"^_ZN3srt13render_kernelILi0E\w+:")      # render_kernel<0, NARROW, ALL_CACHED>: the production builds


def isa_hash():
    """(hash, note) over the bodies of the production render kernels only (label .. s_endpgm of every render_kernel<0,...>): helper
    kernels of the same translation unit (op sweep, scatter, ...) may change without invalidating a counter pass of the renderer.
    None when the listing is missing."""
    if not os.path.exists(ISA):
        return None, "no ISA listing (%s)" % os.path.relpath(ISA, ROOT)
    h = hashlib.sha256()
    inside, n_kernels = False, 0
    for line in open(ISA, errors="replace"):
        if not inside and RENDER.match(line):
            inside, n_kernels = True, n_kernels + 1
        if not inside:
            continue
        body = re.sub(r";.*$", "", line).rstrip()
        if body and not re.match(r"\s*\.(file|ident|loc)\b", body):
            h.update(body.encode() + b"\n")
        if re.match(r"\s*s_endpgm", body):
            inside = False
    if n_kernels == 0:
        return None, "no render_kernel<0,...> in the ISA listing"
    return h.hexdigest(), "sha256 of the gfx950 ISA of the %d production render kernels (comments and .file/.ident/.loc lines removed)" % n_kernels


if __name__ == "__main__":
    print(isa_hash()[0])
