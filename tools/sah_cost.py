#!/usr/bin/env python3
"""SAH cost of a scene's BVH on the CPU (no GPU needed): expected node visits of a random ray that hits the root box."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
sid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene = srt.Scene.builtin(sid, 0).build_bvh(mode, 1984)
left, right, prim, boxes = scene.bvh()
d = boxes[:, 1::2] - boxes[:, 0::2]
area = 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
internal = prim < 0
root = 0
print("nodes", len(prim), "internal", int(internal.sum()), "depth", scene.bvh_depth,
      "SAH internal-visit cost %.3f" % (area[internal].sum() / area[root]), "leaf cost %.3f" % (area[~internal].sum() / area[root]))
