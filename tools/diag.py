#!/usr/bin/env python3
"""GPU diagnostic: instrumented render of a scene, prints per-ray work and wave utilisation."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16); ap.add_argument("--depth", type=int, default=16)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0)
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)
r = srt.Renderer(0)
r.upload_scene(scene)
if a.bvh == 1 and os.environ.get("SRT_TOOL_NO_TUNING") != "1" and srt.pixels_per_lane(r, a.width, a.height, a.world) >= 6.0:
    srt.tune_tree_for_throughput(r, scene, a.width, a.height, a.depth)      # the tree bench.py renders a throughput-bound launch with
r.upload_scene(scene); r.set_camera(cam); r.set_partition(a.rank, a.world)
res = {"tris": scene.n_tris, "nodes": scene.n_nodes, "depth": scene.bvh_depth}
for count in (True, False):
    r.set_count_traversal(count)
    best = 1e30
    for _ in range(a.reps):
        r.init_device_params(a.width, a.height, a.spp, a.depth, 1984)
        r.render_chunk(a.width, a.height)
        r.synchronize()
        best = min(best, r.last_kernel_ms())
    st = r.stats()
    if count:
        u = st["util"]
        res.update(rays=st["rays"], rays_per_path=st["rays"] / st["paths"], V=st["node_visits"] / st["rays"], T=st["tri_tests"] / st["rays"],
                   Bx=st["box_tests"] / st["rays"],
                   trav_simd_util=st["node_visits"] / (64.0 * u[0]), alive_frac=u[1] / (64.0 * u[0]), trav_over_alive=st["node_visits"] / max(u[1], 1), nan_ray_frac=u[2] / st["rays"], fringe_steps_frac=u[3] / max(u[0], 1), fringe_lane_util=u[4] / (64.0 * max(u[3], 1)), inner_lane_util=u[5] / (64.0 * max(u[0] - u[3], 1)), cycles_shade_inner_fringe=[u[6] / max(u[6] + u[7] + u[8], 1), u[7] / max(u[6] + u[7] + u[8], 1), u[8] / max(u[6] + u[7] + u[8], 1)],
                   cyc_per_inner_step=u[7] / max(u[0] - u[3], 1), cyc_per_fringe_step=u[8] / max(u[3], 1), max_pixel_node_visits=st["max_pixel_node_visits"], max_pixel_rays=st["max_pixel_rays"], mean_pixel_node_visits=st["node_visits"] / (a.width * a.height / a.world), ms_instrumented=best,
                   shade_passes=st["shade"][0], lanes_shaded_per_pass=st["shade"][1] / max(st["shade"][0], 1), camera_rays_per_pass=st["shade"][2] / max(st["shade"][0], 1),
                   wave_life_mean_over_max=(st["waves"][1] / max(st["waves"][0], 1)) / max(st["waves"][2], 1), wave_drain_share=st["waves"][3] / max(st["waves"][1], 1), wave_life_max_cycles=st["waves"][2],
                   cyc_per_shade_pass=u[6] / max(st["shade"][0], 1), shade_raw=list(st["shade"]), shade_cycles=u[6], inner_steps=u[0] - u[3], fringe_steps=u[3])
    else:
        res.update(ms=best, mray_s=st["rays"] / best / 1e3)
print(json.dumps(res, indent=1))
