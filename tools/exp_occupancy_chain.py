#!/usr/bin/env python3
"""How fast does a long pixel chain run when fewer waves share its CU?  A frame with fewer tiles than the GPU has wave slots (every
wave gets at most one or two tiles, so the launch time is a chain time) on cfg 5's scene, at 16 / 8 / 4 / 2 / 1 waves per CU
(SRT_WAVES_PER_CU, read when the context is created: one child process per setting).
usage: tools/exp_occupancy_chain.py [--width 480 --height 270 --spp 1024]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def opt(name, d):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
W, H, SPP = opt("--width", 480), opt("--height", 270), opt("--spp", 1024)
CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %r)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
r = srt.Renderer(0)
scene = srt.Scene.builtin(101, 0).build_bvh(1, 1984)
r.upload_scene(scene); r.set_camera(scene.default_camera(W, H)); r.set_partition(0, 1)
ms = []
for _ in range(2):
    r.init_device_params(W, H, spp, 16, 1984); r.render_chunk(W, H); r.synchronize(); ms.append(round(r.last_kernel_ms(), 1))
print("RESULT " + json.dumps({"ms": ms, "rays": r.stats()["rays"]}))
''' % ROOT
for wpc in (16, 8, 4, 2, 1):
    env = dict(os.environ, SRT_WAVES_PER_CU=str(wpc))
    p = subprocess.run([sys.executable, "-c", CHILD, str(W), str(H), str(SPP)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(wpc, "FAILED", p.stderr[-400:], flush=True); continue
    o = json.loads(line[0][7:])
    print("%2d waves per CU (%4d waves for %d tiles): %s ms, %.2f Gray/s" % (wpc, wpc * 256, ((W + 7) // 8) * ((H + 7) // 8), o["ms"], o["rays"] / min(o["ms"]) / 1e6), flush=True)
