#!/usr/bin/env python3
"""Identity of the render kernel's machine code.

Primary: sha256 of the BYTES (PC-relative literals zeroed, see position_independent) of the production render kernels inside the
gfx950 code object of a built libsrt_hip.so -- the
library that is actually loaded (cuda-spectral-ray-tracer_amd.binding.LIB_PATH, i.e. SRT_LIB_PATH when a variant build is being
measured): the .hip_fatbin section holds clang offload bundles, each bundle a gfx950 ELF whose symbol table gives address and size
of every kernel.  `code_hash(lib, narrow, all_cached)`.

Secondary (kept for the round-3 PMC entries, which stored it): sha256 of the kernels' bodies in the ISA listing the build keeps
next to the object (csrc/_build/srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s, from -save-temps), comments and file / ident
directives removed.  A listing says nothing about a library loaded from elsewhere, so `isa_hash` refuses (returns None and why)
when SRT_LIB_PATH is set or the in-tree library is newer than the listing's object.

The PMC passes (tools/pmc_to_lane_ops.py) store the hashes next to the per-ray figures they derive; bench.py recomputes them for
the library it is timing and marks an imported figure whose hash differs as STALE.  Usage: python tools/kernel_id.py [lib]"""
import hashlib
import os
import re
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
OBJ = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels.o")
LIB = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "libsrt_hip.so")


# render_kernel<0, NARROW, ALL_CACHED[, PAIRED]>: the production builds (PAIRED is round 5's fourth template argument; listings / libraries of
# earlier rounds have three)
RENDER = re.compile(r"^_ZN3srt13render_kernelILi0ELb([01])ELb([01])E(?:Lb([01])E)?\w*:")
RENDER_SYM = re.compile(r"^_ZN3srt13render_kernelILi0ELb([01])ELb([01])E(?:Lb([01])E)?EEv\w*$")


def _key(m):
    """(narrow, all_cached) for the general variant, (narrow, all_cached, 1) for the PAIRED one"""
    k = (int(m.group(1)), int(m.group(2)))
    return k + (1,) if m.group(3) == "1" else k
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_function_bytes(elf):
    """{symbol name: bytes} for the FUNC symbols of one ELF64 little-endian image (the gfx950 code object)"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for k in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
        secs.append(dict(type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    out = {}
    for s in secs:
        if s["type"] not in (2, 11) or not s["entsize"]:      # SHT_SYMTAB, SHT_DYNSYM
            continue
        strtab = secs[s["link"]]
        for k in range(s["size"] // s["entsize"]):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, s["off"] + k * s["entsize"])
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= len(secs):      # STT_FUNC, defined
                continue
            end = elf.index(b"\0", strtab["off"] + st_name)
            name = elf[strtab["off"] + st_name:end].decode(errors="replace")
            sec = secs[st_shndx]
            start = sec["off"] + (st_value - sec["addr"])
            out[name] = elf[start:start + st_size]
    return out


def position_independent(code):
    """The kernel's bytes with the PC-relative literals zeroed.  A kernel addresses the constant tables of its code object with
    s_getpc_b64 sN ; s_add_u32 sN, sN, LITERAL ; s_addc_u32 sN+1, sN+1, LITERAL: the literals are distances inside the code object
    and move whenever ANY other function of the translation unit changes size, although not one instruction of this kernel differs
    (measured: three dwords per render kernel).  Zeroing them makes the hash a statement about the kernel's instructions alone."""
    n = len(code) // 4
    w = list(struct.unpack_from("<%dI" % n, code, 0))
    for i in range(n - 2):
        if (w[i] & 0xFF80FFFF) != 0xBE801C00:              # s_getpc_b64 sdst
            continue
        j = i + 1
        if (w[j] & 0xFF80FF00) == 0x8000FF00:               # s_add_u32 sdst, ssrc0, literal
            w[j + 1] = 0
            j += 2
            if j + 1 < n and (w[j] & 0xFF80FF00) == 0x8200FF00:      # s_addc_u32 sdst, ssrc0, literal
                w[j + 1] = 0
    return struct.pack("<%dI" % n, *w) + code[4 * n:]


def code_hashes(lib=LIB):
    """{(narrow, all_cached): sha256 of the kernel's position-independent machine code} for the production render kernels of a built
    library; {} when the file is missing or holds no gfx950 code object with them"""
    out = {}
    try:
        data = open(lib, "rb").read()
    except OSError:
        return out
    pos = data.find(BUNDLE_MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(BUNDLE_MAGIC))
        o = pos + len(BUNDLE_MAGIC) + 8
        for _ in range(min(n, 16)):
            off, size, tlen = struct.unpack_from("<QQQ", data, o)
            triple = data[o + 24:o + 24 + tlen]
            o += 24 + tlen
            if b"gfx950" in triple and size:
                for name, code in _elf_function_bytes(data[pos + off:pos + off + size]).items():
                    m = RENDER_SYM.match(name)
                    if m:
                        out[_key(m)] = hashlib.sha256(position_independent(code)).hexdigest()
        pos = data.find(BUNDLE_MAGIC, pos + 1)
    return out


def variant_name(narrow, all_cached, paired=0):
    return "render_kernel<0,%d,%d%s>" % (narrow, all_cached, ",paired" if paired else "")


def code_hash(lib=LIB, narrow=1, all_cached=1, paired=0):
    """(hash, note) of render_kernel<0, narrow, all_cached[, PAIRED]> inside `lib`"""
    hs = code_hashes(lib)
    key = (narrow, all_cached, 1) if paired else (narrow, all_cached)
    if key not in hs:
        return None, "%s not found in the gfx950 code object of %s" % (variant_name(narrow, all_cached, paired), lib)
    return hs[key], "sha256 of the machine code of %s in the gfx950 code object of %s" % (variant_name(narrow, all_cached, paired), os.path.relpath(lib, ROOT))


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
            h, key = hashlib.sha256(), _key(m)
        body = re.sub(r";.*$", "", line).rstrip()
        if body and not re.match(r"\s*\.(file|ident|loc)\b", body):
            h.update(body.encode() + b"\n")
        if re.match(r"\s*s_endpgm", body):
            out[key] = h.hexdigest()
            h = None
    return out


def isa_hash(narrow=1, all_cached=1, paired=0):
    """(hash, note) of one production variant from the build's listing; (None, why) when the listing cannot speak for the library
    that is loaded: SRT_LIB_PATH names another library, or the in-tree library is older than the listing's object (a listing
    left over from another build)."""
    if os.environ.get("SRT_LIB_PATH"):
        return None, "SRT_LIB_PATH is set: the in-tree ISA listing does not describe the loaded library"
    hs = isa_hashes()
    if not hs:
        return None, "no ISA listing (%s)" % os.path.relpath(ISA, ROOT)
    try:
        if os.path.getmtime(LIB) + 1.0 < os.path.getmtime(OBJ):
            return None, "the ISA listing's object is newer than libsrt_hip.so: the library was not linked from it"
    except OSError:
        pass
    key = (narrow, all_cached, 1) if paired else (narrow, all_cached)
    if key not in hs:
        return None, "%s not in the ISA listing" % variant_name(narrow, all_cached, paired)
    return hs[key], "sha256 of the gfx950 ISA of %s (comments and .file/.ident/.loc lines removed)" % variant_name(narrow, all_cached, paired)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else LIB
    for k, v in sorted(code_hashes(lib).items()):
        print("code    %s %s  (%s)" % (variant_name(*k), v, os.path.relpath(lib, ROOT)))
    for k, v in sorted(isa_hashes().items()):
        print("listing %s %s" % (variant_name(*k), v))
