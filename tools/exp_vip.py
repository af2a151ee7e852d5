#!/usr/bin/env python3
"""A/B of kernel-variant libraries (tools/build_variant.sh) on cfg 5's scene (100k-triangle mesh, inner tree partly served by L2): each
library in a child process (SRT_LIB_PATH) renders (a) a small frame whose framebuffer checksum and ray count must agree across libraries,
(b) one rank's share of a W-rank 3840x2160 frame (the chain-bound case), (c) the whole 1920x1080 frame on one GPU (the throughput case).
usage: tools/exp_vip.py lib1.so lib2.so ... [--spp N] [--world W] [--rank R]   ("default" = the in-tree library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys, zlib
import numpy as np
sys.path.insert(0, %r)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
spp, world, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
r = srt.Renderer(0)
scene = srt.Scene.builtin(101, 0).build_bvh(1, 1984)
out = {}
def run(W, H, s, rk, wd, reps):
    r.upload_scene(scene); r.set_camera(scene.default_camera(W, H)); r.set_partition(rk, wd)
    ms = []
    for _ in range(reps):
        r.init_device_params(W, H, s, 16, 1984)
        r.render_chunk(W, H); r.synchronize()
        ms.append(round(r.last_kernel_ms(), 1))
    cs = 0
    if wd == 1:      # (a share of a W-rank frame is compared by its ray count: the framebuffer needs the other ranks' tiles)
        r.scatter_tiles()
        cs = int(sum(int(p.astype("int64").sum()) for p in r.read_fb()))
    return {"ms": ms, "checksum": cs, "rays": r.stats()["rays"]}
out["small"] = run(640, 360, 64, 0, 1, 1)
out["small_w2"] = run(640, 360, 64, 1, 2, 1)
out["share"] = run(3840, 2160, spp, rank, world, 2)
out["whole"] = run(1920, 1080, 256, 0, 1, 2)
print("RESULT " + json.dumps(out))
''' % ROOT
args = [a for a in sys.argv[1:] if not a.startswith("--")]
def opt(name, d):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
spp, world, rank = opt("--spp", 1024), opt("--world", 8), opt("--rank", 0)
args = [a for a in args if not a.isdigit()]
ref = None
for lib in args:
    env = dict(os.environ)
    if lib != "default":
        env["SRT_LIB_PATH"] = os.path.abspath(lib)
    p = subprocess.run([sys.executable, "-c", CHILD, str(spp), str(world), str(rank)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("%-28s FAILED: %s" % (lib, p.stderr[-600:]), flush=True)
        continue
    o = json.loads(line[0][7:])
    key = [(o[k]["checksum"], o[k]["rays"]) for k in ("small", "small_w2", "share", "whole")]
    if ref is None:
        ref = key
    print("%-28s share of rank %d / %d at %d spp: %s ms   whole 1080p x 256: %s ms   %s" % (
        os.path.basename(lib), rank, world, spp, o["share"]["ms"], o["whole"]["ms"], "exact" if key == ref else "DIFFERENT " + str(key)), flush=True)
