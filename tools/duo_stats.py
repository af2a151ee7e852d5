#!/usr/bin/env python3
"""Event counts of render_kernel_duo (experiment build: tools/build_variant.sh duostats -DSRT_DUO_STATS; run with
SRT_LIB_PATH=gpurun_exp_duostats.so): per ray and per pass, next to the frame time."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=128); ap.add_argument("--depth", type=int, default=16)
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1); r.set_kernel_variant(2)
for _ in range(2):
    r.init_device_params(a.width, a.height, a.spp, a.depth, 1984)
    r.render_chunk(a.width, a.height); r.synchronize()
st = r.stats()
q = [st["node_visits"], st["tri_tests"], st["box_tests"]] + st["util"] + [st["max_pixel_node_visits"], st["max_pixel_rays"]] + st["shade"][:1]
names = ["iter", "swap", "swap_l", "d", "d_l", "g", "g_l", "e", "e_l", "blk", "fr", "fr_l", "asm", "trav_l", "cam_l"]
q = dict(zip(names, q))
rays = st["rays"]
out = {"ms": r.last_kernel_ms(), "mray_s": rays / r.last_kernel_ms() / 1e3, "rays": rays, "launched": r.last_kernel_variant(),
       "rays_per_outer_iteration": rays / max(q["iter"], 1),
       "swap_steps_per_kray": 1e3 * q["swap"] / rays, "lanes_per_swap_step": q["swap_l"] / max(q["swap"], 1),
       "pass_D_per_kray": 1e3 * q["d"] / rays, "lanes_per_pass_D": q["d_l"] / max(q["d"], 1),
       "pass_G_per_kray": 1e3 * q["g"] / rays, "lanes_per_pass_G": q["g_l"] / max(q["g"], 1),
       "pass_E_per_kray": 1e3 * q["e"] / rays, "lanes_per_pass_E": q["e_l"] / max(q["e"], 1), "camera_lanes_per_pass_E": q["cam_l"] / max(q["e"], 1),
       "blocked_lanes_per_service": q["blk"] / max(q["iter"], 1),
       "fringe_steps_per_kray": 1e3 * q["fr"] / rays, "lanes_per_fringe_step": q["fr_l"] / max(q["fr"], 1),
       "traversing_lanes_at_asm_entry": q["trav_l"] / max(q["asm"], 1), "raw": q,
       "env": {k: v for k, v in os.environ.items() if k.startswith("SRT_")}}
print(json.dumps(out, indent=1))
