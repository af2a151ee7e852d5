#!/usr/bin/env python3
"""Strong-scaling emulation on ONE GPU: renders every rank's share of a W-rank job, one after the other, and reports the
render-kernel time of each rank and the max over ranks (= what the W-GPU frame would take, gather aside)."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=1024); ap.add_argument("--depth", type=int, default=16)
ap.add_argument("--worlds", default="1,2,4,8"); ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--ranks", default="", help="only these ranks of every world (default: all)")
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)
r = srt.Renderer(0)
# the trees bench.py renders with: the SAH builder's for chain-bound launches (fewer than 6 pixels per lane), tuned for throughput otherwise
tuned = None
def scene_for(world):
    global tuned
    r.upload_scene(scene)
    if a.bvh != 1 or os.environ.get("SRT_TOOL_NO_TUNING") == "1" or srt.pixels_per_lane(r, a.width, a.height, world) < 6.0:
        return scene
    if tuned is None:
        tuned = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
        srt.tune_tree_for_throughput(r, tuned, a.width, a.height, a.depth)
    return tuned
r.upload_scene(scene); r.set_camera(cam)
r.set_partition(0, 1); r.init_device_params(a.width, a.height, 8, a.depth, 1984); r.render_chunk(a.width, a.height); r.synchronize()   # warm-up
out = {}
for W in [int(x) for x in a.worlds.split(",")]:
    ms, reps = [], []
    r.upload_scene(scene_for(W)); r.set_camera(cam)
    for rank in ([int(x) for x in a.ranks.split(',') if int(x) < W] if a.ranks else range(W)):
        r.set_partition(rank, W)
        best, every = 1e30, []
        for _ in range(a.reps):
            r.init_device_params(a.width, a.height, a.spp, a.depth, 1984)
            r.render_chunk(a.width, a.height); r.synchronize()
            every.append(round(r.last_kernel_ms(), 1))
            best = min(best, r.last_kernel_ms())
        ms.append(round(best, 1)); reps.append(every)
    # per repetition: the frame time is the slowest rank of THAT repetition
    frame = [max(x[k] for x in reps) for k in range(a.reps)]
    out[W] = {"per_rank_ms": ms, "max_ms": max(ms), "speedup_vs_1": None, "per_rank_all_reps": reps, "frame_ms_per_rep": frame,
              "slowest_rank_per_rep": [max(range(len(reps)), key=lambda q: reps[q][k]) for k in range(a.reps)]}
    print("world %d: max-of-best %.1f ms, frames per repetition %s, slowest rank per repetition %s, ranks (best) %s" % (W, max(ms), frame, out[W]["slowest_rank_per_rep"], ms), flush=True)
base = out.get(1, {}).get("max_ms")
for W in out:
    out[W]["speedup_vs_1"] = (base / out[W]["max_ms"]) if base else None
print(json.dumps(out))
