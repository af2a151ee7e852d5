#!/usr/bin/env python3
"""Turns a tools/pmc_passes.sh result (summary.txt + one of its pass*.json bench lines) into the entry of
profiles/r04/lane_ops_per_ray.json that bench.py imports for roofline.achieved.  The entry carries the hash of the kernel's ISA
listing (tools/kernel_id.py) taken from the pass's own bench line, so bench.py can tell a figure that belongs to another binary.

usage: tools/pmc_to_lane_ops.py <pmc dir> <kernel tag> [scene id] [output json] [--traffic <pmc dir of the same workload at another spp>]

Fabric traffic (FETCH_SIZE / WRITE_SIZE: the L2's memory-side requests, Infinity-Cache hits included -- MI355X_MICROARCH.md, HBM section)
is split into a part per PIXEL (RNG state in and out, framebuffer: does not grow with spp) and a part per RAY (tree and shading
records that miss L2).  Round 4 assumed the second part away ("HBM traffic does not depend on spp") -- true for a tree that lives in
LDS / L2 (scene 100: 0.99 GB at 64 spp, 1.03 GB at 1024 spp), false by a factor of 32 for the 100k-triangle mesh (scene 101: 66 B of
raw FETCH per ray).  With --traffic the two parts come from two passes at different spp (two equations per counter); with one pass
everything is booked per ray (an upper bound, said so in the note).  The guide's gfx950 correction -- FETCH_SIZE reports half of the
bytes of 16-byte-per-lane loads -- is applied to the PER-RAY fetch (buffer_load_dwordx4 record loads); the per-pixel part (lone 4-byte
RNG / framebuffer accesses, an uncalibrated pattern) and WRITE_SIZE stay raw."""
import json, os, re, sys
argv = list(sys.argv)
second = None
if "--traffic" in argv:
    k = argv.index("--traffic"); second = argv[k + 1]; del argv[k:k + 2]
key = None
if "--key" in argv:      # entry name (default scene_<id>; e.g. scene_100_builders_tree for another tree of the same scene)
    k = argv.index("--key"); key = argv[k + 1]; del argv[k:k + 2]
sys.argv = argv
d, tag = sys.argv[1], sys.argv[2]
scene = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_id
def read_pass(d):
    c = {}
    for line in open(os.path.join(d, "summary.txt")):
        m = re.match(r"(\S+)\s+per-dispatch\s+(\S+)", line)
        if m:
            c[m.group(1)] = float(m.group(2))
    b = None
    for k in range(1, 12):
        p = os.path.join(d, "pass%d.json" % k)
        if os.path.exists(p) and os.path.getsize(p) > 10:
            b = json.loads(open(p).read().strip().splitlines()[-1]); break
    w, h, spp = [int(x) for x in re.search(r"(\d+)x(\d+), (\d+) spp", b["config"]["workload"]).groups()]
    return c, b, w, h, spp, w * h * spp * b["rays_per_path"]


c, b, w, h, spp, rays = read_pass(d)
pixels = float(w * h)
fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
if second is not None:
    c2, _, w2, h2, spp2, rays2 = read_pass(second)
    assert (w2, h2) == (w, h) and spp2 != spp, "--traffic needs the same frame at another spp"
    f2, wr2 = c2["FETCH_SIZE"] * 1024.0, c2["WRITE_SIZE"] * 1024.0
    bf = (f2 - fetch) / (rays2 - rays); af = (fetch - bf * rays) / pixels
    bw = (wr2 - write) / (rays2 - rays); aw = (write - bw * rays) / pixels
    bf, bw, af, aw = max(bf, 0.0), max(bw, 0.0), max(af, 0.0), max(aw, 0.0)
    fabric_note = ("two-point split of FETCH_SIZE / WRITE_SIZE (separate --pmc passes at %d and %d spp): per pixel %.1f B fetched + %.1f B written (raw), "
                   "per ray %.3f B fetched raw -> x 2 (gfx950: 16-byte-per-lane loads are tallied at half) + %.3f B written" % (spp, spp2, af, aw, bf, bw))
else:
    af = aw = 0.0
    bf, bw = fetch / rays, write / rays
    fabric_note = "ONE pass (%d spp): everything booked per ray (upper bound for the per-ray part; the per-pixel part is in it); fetch x 2 (gfx950 correction for 16-byte-per-lane loads)" % spp
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
    "fabric_bytes_per_pixel": af + aw, "fabric_bytes_per_ray": 2.0 * bf + bw, "fabric_note": fabric_note,
    "fabric_raw": {"fetch_bytes": fetch, "write_bytes": write, "pixels": pixels, "rays": rays, "spp": spp},
    "counters": c,
}
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r05", "lane_ops_per_ray.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
doc = json.load(open(out)) if os.path.exists(out) else {}
doc[key or "scene_%d" % scene] = entry
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in entry.items() if k != "counters"}, indent=1))
