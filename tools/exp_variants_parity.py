#!/usr/bin/env python3
"""A/B of kernel-variant libraries (tools/build_variant.sh) on one box: each library in a child process (SRT_LIB_PATH) renders a few
small images + the timed workload; the parent compares framebuffer / XYZ checksums across libraries and prints the times.
usage: tools/exp_variants_parity.py lib1.so lib2.so ... [--spp N]   ("default" = the in-tree library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys, zlib
import numpy as np
sys.path.insert(0, %r)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
r = srt.Renderer(0)
out = {"small": []}
for sid, bvh, W, H, spp, depth in ((1, 0, 64, 64, 8, 8), (100, 1, 200, 120, 12, 16), (0, 0, 97, 61, 9, 8), (2, 0, 120, 80, 6, 16), (100, 1, 640, 360, 16, 16), (1, 0, 320, 180, 64, 16)):
    scene = srt.Scene.builtin(sid, 0).build_bvh(bvh, 1984)
    img = srt.render_image(scene, scene.default_camera(W, H), W, H, spp, depth, renderer=r)
    out["small"].append([zlib.crc32(np.concatenate(img["xyz"]).tobytes()), zlib.crc32(np.concatenate(img["fb"]).tobytes()), img["stats"]["rays"]])
spp = int(sys.argv[1])
for name, sid, bvh, W, H, s in (("cfg3", 100, 1, 1920, 1080, spp), ("cfg2", 100, 1, 1280, 720, 256), ("cfg4", 1, 0, 1920, 1080, max(spp // 2, 64))):
    scene = srt.Scene.builtin(sid, 0).build_bvh(bvh, 1984)
    cam = scene.default_camera(W, H)
    r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
    best = 1e30
    for _ in range(3):
        r.init_device_params(W, H, s, 16, 1984)
        r.render_chunk(W, H); r.synchronize()
        best = min(best, r.last_kernel_ms())
    r.scatter_tiles()
    out[name] = {"ms": best, "checksum": int(sum(int(p.astype("int64").sum()) for p in r.read_fb())), "spp": s}
print("RESULT " + json.dumps(out))
''' % ROOT
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 1024
res = {}
for lib in libs:
    env = dict(os.environ)
    if lib != "default":
        env["SRT_LIB_PATH"] = os.path.abspath(lib)
    else:
        env.pop("SRT_LIB_PATH", None)
    p = subprocess.run([sys.executable, "-c", CHILD, str(spp)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(lib, "FAILED", p.stderr[-800:]); continue
    res[lib] = json.loads(line[0][7:])
base = res.get(libs[0])
for lib, r in res.items():
    same = base is not None and r["small"] == base["small"] and all(r[k]["checksum"] == base[k]["checksum"] for k in ("cfg3", "cfg2", "cfg4"))
    print("%-28s cfg3 %.2f ms  cfg2 %.2f ms  cfg4 %.2f ms   images %s" % (lib, r["cfg3"]["ms"], r["cfg2"]["ms"], r["cfg4"]["ms"], "== first library" if same else "DIFFER from the first library"), flush=True)
