#!/usr/bin/env python3
"""Where the spilled SGPRs of the headline's production kernel are touched: counts v_readlane / v_writelane in render_kernel<0,1,1> of
the build's ISA listing by loop level -- the innermost loop around the assembly block (.Lsrt_phase_decide .. .Lsrt_phase_end: scheduling
decision + INNER bursts), the traversal phase (that loop + the FRINGE visit it returns to), the persistent outer loop, the rest.
No GPU needed.  usage: python tools/spill_report.py"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "srt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
RU = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc", "_build", "resource_usage.txt")
lines = open(S, errors="replace").read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN3srt13render_kernelILi0ELb1ELb1E\w*:", l)][0]
end = next(i for i in range(start, len(lines)) if re.match(r"\s*s_endpgm", lines[i]))
body = lines[start:end]
rl = [i for i, l in enumerate(body) if "v_readlane" in l or "v_writelane" in l]
a = next(i for i, l in enumerate(body) if "Lsrt_phase_decide" in l and l.strip().endswith(":"))
b = next(i for i, l in enumerate(body) if "Lsrt_phase_end" in l and l.strip().endswith(":"))
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
back = []
for i, l in enumerate(body):
    m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        back.append((labels[m.group(1)], i))
enc = sorted([(s, e) for s, e in back if s <= a and e >= b], key=lambda t: t[1] - t[0])
spilled = None
blk = open(RU, errors="replace").read()
m = re.search(r"render_kernelILi0ELb1ELb1E.*?SGPRs Spill: (\d+)", blk, re.S)
if m:
    spilled = int(m.group(1))
print("render_kernel<0,1,1>: %s spilled SGPRs, %d v_readlane / v_writelane instructions in %d lines of ISA" % (spilled, len(rl), len(body)))
names = ["innermost loop around the assembly block (decision + INNER bursts)", "traversal phase (+ the FRINGE visit)", "persistent outer loop (shading pass, pixel switch, camera ray, traversal)"]
seen = 0
for (s, e), name in zip(enc[:1] + enc[1:2] + enc[-1:], names):
    n = sum(1 for i in rl if s <= i <= e)
    print("  %-78s lines %5d..%-5d  spill traffic %d" % (name, s, e, n))
print("  %-78s %s  spill traffic %d" % ("prologue / epilogue", " " * 18, len(rl) - sum(1 for i in rl if enc[-1][0] <= i <= enc[-1][1])))
