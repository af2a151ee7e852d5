#!/usr/bin/env python3
"""Census of long pixel chains in one rank's share of a frame: the cost probe's most expensive pixel per tile (node visits, scaled to
the frame's samples) against the mean load of a lane (all of the rank's probe cost / the lanes of the launch).  A pixel whose chain is
longer than a lane's mean load bounds the launch whatever the queue order.  Run with SRT_PROBE_SPP=32 for a low-noise probe.
usage: SRT_PROBE_SPP=32 tools/chain_census.py [--scene 101 --width 3840 --height 2160 --world 8 --rank 0]"""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=101); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
a = ap.parse_args()
probe = int(os.environ.get("SRT_PROBE_SPP", "2"))
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(scene.default_camera(a.width, a.height)); r.set_partition(a.rank, a.world)
r.init_device_params(a.width, a.height, 4 * probe + 1, 16, 1984); r.render_chunk(a.width, a.height); r.synchronize()
cost, mx = r.tile_costs(with_max_pixel=True)
cost, mx = cost.astype(np.float64), mx.astype(np.float64)
plan = r.launch_plan()
n_lanes = 256 * 64 * int(plan.get("waves_per_cu", 16)) if isinstance(plan, dict) else 256 * 64 * 16
lane_load = cost.sum() / n_lanes                      # mean probe cost a lane carries (same unit as mx: node visits at `probe` samples)
ratio = mx / lane_load
out = {"rank": a.rank, "world": a.world, "tiles": int(cost.size), "probe_spp": probe, "lanes": n_lanes, "pixels_per_lane": cost.size * 64 / n_lanes,
       "longest_pixel_over_lane_load": float(ratio.max()), "tiles_with_a_pixel_over_x_lane_loads": {str(x): int((ratio > x).sum()) for x in (0.4, 0.5, 0.6, 0.7, 0.8, 1.0, 1.2)},
       "share_of_cost_in_those_tiles": {str(x): float(cost[ratio > x].sum() / cost.sum()) for x in (0.4, 0.5, 0.6, 0.7, 0.8, 1.0, 1.2)},
       "tile_cost_over_mean_percentiles_50_90_99_100": [float(v) for v in np.percentile(cost / cost.mean(), [50, 90, 99, 100])]}
print(json.dumps(out))
