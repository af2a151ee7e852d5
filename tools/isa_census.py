#!/usr/bin/env python3
"""Instruction census of one render_kernel variant from the build's gfx950 listing (csrc/_build/*.s): one line per basic block --
instructions, of which vector ALU / scalar / LDS / vector memory, the branches that leave it and the compiler's block name -- and the
totals of the loops.  Used for the shading-pass budget in DESIGN.md 5.3 (round 5, third session).
usage: tools/isa_census.py [mangled-substring, default ILi0ELb1ELb1ELb1 = render_kernel<0,1,1,paired>]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
want = sys.argv[1] if len(sys.argv) > 1 else "ILi0ELb1ELb1ELb1"
lines = open(S).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3srt13render_kernel%s\w*:" % want, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
def new(name, note, i):
    global cur
    cur = {"name": name, "note": note, "line": i - start, "n": 0, "v": 0, "s": 0, "ds": 0, "vm": 0, "br": []}
    blocks.append(cur)
new("entry", "", start)
for i in range(start + 1, end):
    s = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", s)
    if m:
        new(m.group(1), (m.group(2) or "").lstrip("; "), i); continue
    m = re.match(r"^; %bb\.(\d+):\s*(;.*)?", s)
    if m:
        new("bb." + m.group(1), (m.group(2) or "").lstrip("; "), i); continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    cur["n"] += 1
    if op.startswith("v_"): cur["v"] += 1
    elif op.startswith("s_"): cur["s"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.split("_")[0] in ("buffer", "global", "flat"): cur["vm"] += 1
    if op.startswith("s_cbranch") or op == "s_branch":
        cur["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + ">" + re.sub(r"^\.LBB\d+_", "", s.split()[1]))
tot = {"n": 0, "v": 0, "s": 0, "ds": 0, "vm": 0}
for b in blocks:
    for k in tot: tot[k] += b[k]
    print("%-10s +%-5d n=%-4d v=%-4d s=%-4d ds=%-3d vm=%-3d %-28s %s" % (re.sub(r"^\.LBB\d+_", "B", b["name"]), b["line"], b["n"], b["v"], b["s"], b["ds"], b["vm"],
                                                                     " ".join(b["br"])[:28], b["note"][:90]))
print("total: %(n)d instructions, %(v)d vector ALU, %(s)d scalar, %(ds)d LDS, %(vm)d vector memory" % tot)
