#!/usr/bin/env python3
"""What order_tiles_kernel's split policy decides for rank 0 of a W-rank job (numpy replica of the bisection; diagnostics)."""
import argparse, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=4); a = ap.parse_args()
W, H, depth = 1920, 1080, 16
scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
cam = scene.default_camera(W, H)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, a.world)
r.init_device_params(W, H, 32, depth, 1984)
r.render_chunk(W, H); r.synchronize()
cost = r.tile_costs().astype(np.float64)
g = np.array([1.0, 0.957, 0.863, 0.794, 0.767, 0.687, 0.442])
n_waves = 4096
def levels(T):
    s = np.zeros(cost.size, dtype=int)
    for k in range(6):
        s = np.where((s == k) & (cost * g[k] > T), k + 1, s)
    return s
print("tiles", cost.size, "mean", cost.mean(), "max/mean %.2f" % (cost.max() / cost.mean()), "sum/n_waves/mean %.2f" % (cost.sum() / n_waves / cost.mean()))
for load_pct in (100, 150, 220):
    lf = load_pct / 100
    lo, hi = cost.max() * g[6], max(cost.max(), lf * cost.sum() / n_waves)
    for it in range(14):
        mid = 0.5 * (lo + hi); s = levels(mid)
        load = ((2.0 ** s) * cost * g[s]).sum()
        if lf * load <= n_waves * mid: hi = mid
        else: lo = mid
    s = levels(hi)
    print("load %d: target/mean %.2f rows %d, tiles per level %s, crit rows %d" % (load_pct, hi / cost.mean(), int((2 ** s).sum()), np.bincount(s, minlength=7).tolist(),
          int(((2 ** s) * ((cost * g[s] >= 0.5 * hi) & (hi < 1.5 * cost.max()))).sum())))
