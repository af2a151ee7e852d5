#!/usr/bin/env python3
"""Turns a tools/pmc_passes.sh result (summary.txt + one of its pass*.json bench lines) into the entry of
profiles/r04/lane_ops_per_ray.json that bench.py imports for roofline.achieved.  The entry carries the hash of the kernel's ISA
listing (tools/kernel_id.py) taken from the pass's own bench line, so bench.py can tell a figure that belongs to another binary.

usage: tools/pmc_to_lane_ops.py <pmc dir> <kernel tag> [scene id] [output json]
The PMC passes render the headline scene at 64 spp; per-ray figures do not depend on spp, HBM traffic does not either
(it is RNG state + framebuffer per pixel: 64-spp and 1024-spp launches read the same FETCH_SIZE / WRITE_SIZE)."""
import json, os, re, sys
d, tag = sys.argv[1], sys.argv[2]
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_id
c = {}
for line in open(os.path.join(d, "summary.txt")):
    m = re.match(r"(\S+)\s+per-dispatch\s+(\S+)", line)
    if m:
        c[m.group(1)] = float(m.group(2))
b = None
for k in range(1, 9):
    p = os.path.join(d, "pass%d.json" % k)
    if os.path.exists(p) and os.path.getsize(p) > 10:
        b = json.loads(open(p).read().strip().splitlines()[-1]); break
w, h, spp = [int(x) for x in re.search(r"(\d+)x(\d+), (\d+) spp", b["config"]["workload"]).groups()]
rays = w * h * spp * b["rays_per_path"]
wc = c["SQ_WAVE_CYCLES"]
entry = {
    "kernel": tag, "kernel_code_sha256": (b.get("roofline") or {}).get("kernel_code_sha256"), "kernel_isa_sha256": (b.get("roofline") or {}).get("kernel_isa_sha256"), "tree_sha256": (b.get("roofline") or {}).get("tree_sha256"), "kernel_variant": (b.get("roofline") or {}).get("kernel"), "source": "tools/pmc_passes.sh: rocprofv3 --pmc <set> --kernel-trace, one counter set per run, render_kernel<0,...> dispatch, %dx%d, %d spp" % (w, h, spp),
    "rays_in_pmc_launch": rays,
    "lane_ops_per_ray": c["SQ_THREAD_CYCLES_VALU"] / rays,
    "valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / rays, "salu_wave_instr_per_ray": c["SQ_INSTS_SALU"] / rays,
    "lds_wave_instr_per_ray": c["SQ_INSTS_LDS"] / rays, "vmem_wave_instr_per_ray": (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / rays,
    "lanes_per_valu_instruction": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
    "wave_time_split": {"active": c["SQ_ACTIVE_INST_ANY"] / wc, "wait_inst_issue": c["SQ_WAIT_INST_ANY"] / wc, "wait_waitcnt": c["SQ_WAIT_ANY"] / wc},
    "instr_per_cycle_per_simd": (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"] + c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0),
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "lds_bank_conflict_cycles_per_lds_instr": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"],
    "hbm_bytes_per_launch_1024spp": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "hbm_note": "FETCH_SIZE + WRITE_SIZE (KiB) of the dispatch, raw; the gfx950 2x correction of MI355X_MICROARCH.md applies to wide coalesced streams, these are lone 4-byte RNG / result accesses (uncalibrated pattern), so the raw value is kept",
    "counters": c,
}
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r04", "lane_ops_per_ray.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
doc = json.load(open(out)) if os.path.exists(out) else {}
doc["scene_%d" % scene] = entry
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in entry.items() if k != "counters"}, indent=1))
