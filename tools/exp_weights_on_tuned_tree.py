#!/usr/bin/env python3
"""Step-choice weights (SRT_SCORE_SHADE / SRT_SCORE_FRINGE, read when a context is created) re-swept on the throughput-tuned tree of
cfg 3's scene: one child process per setting, 1920x1080 x 256 spp, best of 3 frames.  usage: tools/exp_weights_on_tuned_tree.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %r)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
r = srt.Renderer(0)
scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
W, H = 1920, 1080
r.upload_scene(scene)
srt.tune_tree_for_throughput(r, scene, W, H, 16)
r.upload_scene(scene); r.set_camera(scene.default_camera(W, H)); r.set_partition(0, 1)
ms = []
for _ in range(3):
    r.init_device_params(W, H, 256, 16, 1984); r.render_chunk(W, H); r.synchronize(); ms.append(round(r.last_kernel_ms(), 2))
print("RESULT " + json.dumps(ms))
''' % ROOT
for sh in (55, 70, 85, 100):
    for fr in (220, 280, 340, 420):
        env = dict(os.environ, SRT_SCORE_SHADE=str(sh), SRT_SCORE_FRINGE=str(fr))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print("shade %3d fringe %3d: %s" % (sh, fr, line[0][7:] if line else "FAILED " + p.stderr[-200:]), flush=True)
