#!/usr/bin/env python3
"""render_kernel_duo against render_kernel (and so against the oracle, which the GPU suite holds render_kernel to): bit-identical
planes on a few scenes / sizes, then the frame time of both on a workload.  Usage: tools/duo_check.py [--time W H SPP] [--scene N]"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--time", nargs=3, type=int, default=None, metavar=("W", "H", "SPP"))
ap.add_argument("--depth", type=int, default=16); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--skip-parity", action="store_true")
a = ap.parse_args()
r = srt.Renderer(0)
bad = 0
if not a.skip_parity:
    for sid, bvh, W, H, spp, depth in ((1, 0, 64, 64, 8, 8), (100, 1, 200, 120, 12, 16), (0, 0, 97, 61, 9, 8), (2, 0, 120, 80, 6, 16), (100, 1, 333, 187, 5, 3), (1, 0, 50, 30, 4, 0), (1, 0, 50, 30, 4, 1),
                                      (100, 1, 640, 360, 16, 16)):
        scene = srt.Scene.builtin(sid, 0).build_bvh(bvh, 1984)
        cam = scene.default_camera(W, H)
        one = srt.render_image(scene, cam, W, H, spp, depth, renderer=r, variant=1)
        two = srt.render_image(scene, cam, W, H, spp, depth, renderer=r, variant=2)
        diff = {k: int(sum(int(np.count_nonzero(x.view(np.uint32) != y.view(np.uint32))) for x, y in zip(one[k], two[k]))) for k in ("fb", "lin", "xyz")}
        ok = all(v == 0 for v in diff.values()) and one["stats"]["rays"] == two["stats"]["rays"] and two["variant"] == 1 and one["variant"] == 0
        bad += 0 if ok else 1
        print("scene %3d %4dx%-4d %3d spp depth %2d: variants %d/%d rays %d/%d differing lanes %s %s" %
              (sid, W, H, spp, depth, one["variant"], two["variant"], one["stats"]["rays"], two["stats"]["rays"], diff, "OK" if ok else "MISMATCH"), flush=True)
if a.time:
    W, H, spp = a.time
    scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
    cam = scene.default_camera(W, H)
    r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
    res = {}
    for variant in (1, 2):
        r.set_kernel_variant(variant)
        best = 1e30
        for _ in range(a.reps):
            r.init_device_params(W, H, spp, a.depth, 1984)
            r.render_chunk(W, H); r.synchronize()
            best = min(best, r.last_kernel_ms())
        rays = r.stats()["rays"]
        r.scatter_tiles()
        fb = r.read_fb()
        res[variant] = dict(ms=best, mray_s=rays / best / 1e3, launched=r.last_kernel_variant(), checksum=int(sum(int(p.astype("int64").sum()) for p in fb)))
    print(json.dumps(dict(workload="scene %d %dx%d %d spp" % (a.scene, W, H, spp), render_kernel=res[1], render_kernel_duo=res[2],
                          speedup=res[1]["ms"] / res[2]["ms"], env={k: v for k, v in os.environ.items() if k.startswith("SRT_")})), flush=True)
sys.exit(1 if bad else 0)
