#!/usr/bin/env python3
"""Identity of the render kernel's machine code: sha256 of the gfx950 ISA listing the build keeps next to the object
(csrc/_build/srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s, from -save-temps), comments and file / ident directives removed.

The PMC passes (tools/pmc_to_lane_ops.py) store it next to the per-ray figures they derive; bench.py recomputes it for the
library it is timing and marks an imported figure whose hash differs as STALE.  Usage: python tools/kernel_id.py"""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
LIB = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "libsrt_hip.so")


def isa_hash():
    """(hash, note).  None when the listing is missing or older than the library (a library built some other way)."""
    if not os.path.exists(ISA):
        return None, "no ISA listing (%s)" % os.path.relpath(ISA, ROOT)
    h = hashlib.sha256()
    for line in open(ISA, errors="replace"):
        line = re.sub(r";.*$", "", line).rstrip()
        if not line or re.match(r"\s*\.(file|ident|loc)\b", line):
            continue
        h.update(line.encode() + b"\n")
    return h.hexdigest(), "sha256 of the gfx950 ISA listing of srt_kernels.hip (comments and .file/.ident/.loc lines removed)"


if __name__ == "__main__":
    print(isa_hash()[0])
