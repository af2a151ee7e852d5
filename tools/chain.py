#!/usr/bin/env python3
"""Latency floor: find the most expensive 8x8 tile with the probe, then render tiles around it ALONE at full spp."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
W, H, spp, depth = 1920, 1080, 1024, 16
scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
cam = scene.default_camera(W, H)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
r.init_device_params(W, H, 32, depth, 1984)
r.render_chunk(W, H); r.synchronize()
cost = r.tile_costs().astype(np.float64)
tiles_x = (28 * (W // 28 + 1) + 7) // 8      # tiles cover the whole reference grid (srt_render_chunk)
order = np.argsort(-cost)
print("tiles", cost.size, "mean", cost.mean(), "max/mean", cost.max() / cost.mean(), "top10/mean", (cost[order[:10]] / cost.mean()).round(1).tolist())
for q in (50, 90, 99, 99.9):
    print("percentile", q, "cost/mean = %.2f" % (np.percentile(cost, q) / cost.mean()))
for frac in (2, 4, 8):
    print("tiles >= %dx mean: %d" % (frac, int((cost >= frac * cost.mean()).sum())))
t = int(order[0]); tx, ty = t % tiles_x, t // tiles_x
print("most expensive tile at pixel", tx * 8, ty * 8)
for (cw, ch) in ((8, 8), (64, 64)):
    ox, oy = max(0, tx * 8 - (cw - 8) // 2), max(0, ty * 8 - (ch - 8) // 2)
    r.init_device_params(cw, ch, spp, depth, 1984)
    r.set_count_traversal(True)
    r.render_chunk(cw, ch, ox, oy); r.synchronize()
    st = r.stats(); ms = r.last_kernel_ms()
    print("chunk %dx%d alone: %.1f ms, rays %d, max_pixel_node_visits %d max_pixel_rays %d -> %.2f us per node visit of the worst pixel" %
          (cw, ch, ms, st["rays"], st["max_pixel_node_visits"], st["max_pixel_rays"], ms * 1e3 / max(st["max_pixel_node_visits"], 1)))
    u = st["util"]
    n_in, n_fr = u[0] - u[3], u[3]
    print("   wave steps: inner %d (%.0f cyc each, %.1f lanes), fringe %d (%.0f cyc each, %.1f lanes); shading cycles total %d; node_visits %d; V %.1f T %.1f" %
          (n_in, u[7] / max(n_in, 1), u[5] / max(n_in, 1), n_fr, u[8] / max(n_fr, 1), u[4] / max(n_fr, 1), u[6], st["node_visits"],
           st["node_visits"] / st["rays"], st["tri_tests"] / st["rays"]))
