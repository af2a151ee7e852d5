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


RENDER = re.compile(r"^_ZN3srt13render_kernelILi0ELb([01])ELb([01])E\w*:")      # render_kernel<0, NARROW, ALL_CACHED>: the production builds


def isa_hashes():
    """{(narrow, all_cached): sha256} over the body (label .. s_endpgm) of every production render kernel in the listing: helper
    kernels of the same translation unit (op sweep, scatter, ...) and the OTHER variants may change without invalidating a counter
    pass of one variant.  Empty when the listing is missing."""
    out = {}
    if not os.path.exists(ISA):
        return out
    h, key = None, None
    for line in open(ISA, errors="replace"):
        if h is None:
            m = RENDER.match(line)
            if not m:
                continue
            h, key = hashlib.sha256(), (int(m.group(1)), int(m.group(2)))
        body = re.sub(r";.*$", "", line).rstrip()
        if body and not re.match(r"\s*\.(file|ident|loc)\b", body):
            h.update(body.encode() + b"\n")
        if re.match(r"\s*s_endpgm", body):
            out[key] = h.hexdigest()
            h = None
    return out


def isa_hash(narrow=1, all_cached=1):
    """(hash, note) of one production variant: render_kernel<0, narrow, all_cached> (default: the headline's)."""
    hs = isa_hashes()
    if not hs:
        return None, "no ISA listing (%s)" % os.path.relpath(ISA, ROOT)
    if (narrow, all_cached) not in hs:
        return None, "render_kernel<0,%d,%d> not in the ISA listing" % (narrow, all_cached)
    return hs[(narrow, all_cached)], "sha256 of the gfx950 ISA of render_kernel<0,%d,%d> (comments and .file/.ident/.loc lines removed)" % (narrow, all_cached)


if __name__ == "__main__":
    for k, v in sorted(isa_hashes().items()):
        print("render_kernel<0,%d,%d> %s" % (k[0], k[1], v))
