#!/usr/bin/env python3
"""Render ONE 8x8 tile alone at full spp (latency experiments; SRT_DEBUG_LANE_LIMIT=p keeps only the first p pixels)."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--x", type=int, default=1232); ap.add_argument("--y", type=int, default=720)
ap.add_argument("--spp", type=int, default=1024); ap.add_argument("--count", type=int, default=0)
a = ap.parse_args()
W, H, depth = 1920, 1080, 16
scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
cam = scene.default_camera(W, H)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
best = 1e30
for rep in range(2):
    r.init_device_params(8, 8, a.spp, depth, 1984)
    r.set_count_traversal(bool(a.count))
    r.render_chunk(8, 8, a.x, a.y); r.synchronize()
    best = min(best, r.last_kernel_ms())
st = r.stats()
print("lane_limit", os.environ.get("SRT_DEBUG_LANE_LIMIT", "64"), "ms %.1f" % best, "rays", st["rays"], "max_pixel_node_visits", st.get("max_pixel_node_visits"))
print("  shade counters", list(st["shade"]))
if a.count:
    u = st["util"]; n_in, n_fr = u[0] - u[3], u[3]
    tot = max(u[6] + u[7] + u[8], 1)
    print("  instrumented: inner steps %d (%.0f cyc each), fringe steps %d (%.0f cyc each), shading passes %d (%.0f cyc each); shares shade/inner/fringe %.2f %.2f %.2f; cycles per ray %.0f" %
          (n_in, u[7] / max(n_in, 1), n_fr, u[8] / max(n_fr, 1), st["shade"][0], u[6] / max(st["shade"][0], 1), u[6] / tot, u[7] / tot, u[8] / tot, tot / max(st["rays"], 1)))
