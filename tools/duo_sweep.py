#!/usr/bin/env python3
"""Knob sweep of render_kernel_duo in one process: the scheduling weights / fill thresholds are read from the environment when a
context is created, so every setting gets its own context.  Prints one line per setting, best first at the end.
usage: tools/duo_sweep.py [--spp N] [--grid 'W_SWAP=256,1024;W_BLOCKED=70,280;FILL_D=24,40;FILL_E=24,40;FILL_G=8,24']"""
import argparse, importlib, itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=128); ap.add_argument("--depth", type=int, default=16); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--grid", default="W_SWAP=256,1024,4096;W_BLOCKED=70,280,1120;FILL_D=16,32,48;FILL_E=16,32,48;FILL_G=8,24")
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)


def run(env, variant):
    for k in list(os.environ):
        if k.startswith("SRT_DUO_"):
            del os.environ[k]
    os.environ.update(env)
    r = srt.Renderer(0)
    r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1); r.set_kernel_variant(variant)
    best = 1e30
    for _ in range(a.reps):
        r.init_device_params(a.width, a.height, a.spp, a.depth, 1984)
        r.render_chunk(a.width, a.height); r.synchronize()
        best = min(best, r.last_kernel_ms())
    r.close()
    return best


base = run({}, 1)
print("render_kernel: %.2f ms" % base, flush=True)
axes = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in a.grid.split(";") if kv]
res = []
for combo in itertools.product(*[v for _, v in axes]):
    env = {"SRT_DUO_" + k: v for (k, _), v in zip(axes, combo)}
    ms = run(env, 2)
    res.append((ms, env))
    print("%.2f ms (%.3fx)  %s" % (ms, base / ms, " ".join("%s=%s" % (k[8:], v) for k, v in env.items())), flush=True)
res.sort(key=lambda t: t[0])
print("best:")
for ms, env in res[:8]:
    print("  %.2f ms (%.3fx)  %s" % (ms, base / ms, " ".join("%s=%s" % (k[8:], v) for k, v in env.items())))
