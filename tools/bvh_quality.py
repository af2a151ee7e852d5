#!/usr/bin/env python3
"""V / T per ray of a scene's BVH measured with the CPU checker (tools only; no GPU needed).  Note: the checker walks
NaN-direction rays through the whole tree like the reference, so V here includes them (the kernel's shortcut does not)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
import oracle_binding as O
from helpers import oracle_scene_for
sid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H, spp, depth = 240, 135, 4, 16
scene = srt.Scene.builtin(sid, 0).build_bvh(1, 1984)
cam = scene.default_camera(W, H)
osc = oracle_scene_for(O, scene, 1)
t0 = time.time()
img = osc.render(cam, W, H, spp, depth, threads=8)
st = img["stats"]
print("scene", sid, "rays/path %.3f V %.2f T %.2f box %.2f max_stack %d (%.1f s)" % (st["rays"] / st["paths"], st["trav_iters"] / st["rays"],
      st["tri_tests"] / st["rays"], st["box_tests"] / st["rays"], st["max_stack"], time.time() - t0))
