#!/usr/bin/env python3
"""The tail of a launch: per-wave life times of an instrumented render (srt_get_wave_debug).  Prints the distribution of wave end
times, what the longest-lived waves were doing (rays, most expensive pixel) and the share of wave-slot time lost after exit."""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
ap.add_argument("--spp", type=int, default=256); ap.add_argument("--depth", type=int, default=16)
ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0)      # one rank's share of a W-rank job
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)
r = srt.Renderer(0)
r.upload_scene(scene)
if a.bvh == 1 and os.environ.get("SRT_TOOL_NO_TUNING") != "1" and srt.pixels_per_lane(r, a.width, a.height, a.world) >= 6.0:
    srt.tune_tree_for_throughput(r, scene, a.width, a.height, a.depth)      # the tree bench.py renders a throughput-bound launch with
r.upload_scene(scene); r.set_camera(cam); r.set_partition(a.rank, a.world)
r.set_count_traversal(True)
for _ in range(a.reps):
    r.init_device_params(a.width, a.height, a.spp, a.depth, 1984)
    r.render_chunk(a.width, a.height); r.synchronize()
w = r.wave_debug().astype(np.float64)
life, dry, rays, maxpix = w[:, 0], w[:, 1], w[:, 2], w[:, 3]
mx = life.max()
st = r.stats()
u = st["util"]
# where the wave-slot time of the launch goes: [born .. queue dry] steady state, [dry .. own end] drain with emptying waves, [own end .. launch end] idle slot
n = w.shape[0]
dry_c = np.where(dry < 4e9, np.minimum(dry, life), life)
print(json.dumps({"slot_time_shares": {"steady_until_queue_dry": float(dry_c.sum() / (n * mx)), "drain_after_queue_dry": float((life - dry_c).sum() / (n * mx)),
                                       "idle_after_wave_exit": float((mx - life).sum() / (n * mx))},
                  "rank": a.rank, "world": a.world, "rays": int(st["rays"]),
                  "inner_lane_util": u[5] / (64.0 * max(u[0] - u[3], 1)), "fringe_lane_util": u[4] / (64.0 * max(u[3], 1)),
                  "lanes_shaded_per_pass": st["shade"][1] / max(st["shade"][0], 1),
                  "cycles_shade_inner_fringe": [u[6] / max(u[6] + u[7] + u[8], 1), u[7] / max(u[6] + u[7] + u[8], 1), u[8] / max(u[6] + u[7] + u[8], 1)]}))
q = np.percentile(life / mx, [0, 5, 25, 50, 75, 95, 99, 100])
print(json.dumps({"waves": int(w.shape[0]), "kernel_ms": r.last_kernel_ms(), "life_over_max_percentiles_0_5_25_50_75_95_99_100": [round(float(x), 3) for x in q],
                  "mean_life_over_max": float(life.mean() / mx), "mean_dry_over_max": float(dry[dry < 4e9].mean() / mx),
                  "rays_per_wave_mean": float(rays.mean()), "rays_per_wave_p99": float(np.percentile(rays, 99))}))
order = np.argsort(-life)[:12]
print("longest-lived waves: life/max, dry/max, rays (vs mean), most expensive pixel's rays")
for k in order:
    print("  %.3f  %.3f  %8d (%.2fx)  %6d" % (life[k] / mx, min(dry[k], 4e9) / mx, rays[k], rays[k] / rays.mean(), maxpix[k]))
order = np.argsort(life)[:6]
print("shortest-lived waves:")
for k in order:
    print("  %.3f  %.3f  %8d (%.2fx)  %6d" % (life[k] / mx, min(dry[k], 4e9) / mx, rays[k], rays[k] / rays.mean(), maxpix[k]))
# correlation of a wave's life with its most expensive pixel
print("corr(life, max pixel rays) = %.3f, corr(life, rays) = %.3f" % (np.corrcoef(life, maxpix)[0, 1], np.corrcoef(life, rays)[0, 1]))
