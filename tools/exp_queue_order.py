#!/usr/bin/env python3
"""cfg 5's scene (100k-triangle mesh, tree served by L2): does the ORDER of the pixel queue matter for the L2 hit rate?  The default queue is
cost-descending (longest-processing-time-first), which scatters the tiles that are in flight together over the whole image; SRT_PROBE_SPP=0
hands the tiles out in raster order (a band of ~70 pixel rows in flight at 3840x2160), the other settings move the sort key.  Each setting
in a child process (the knobs are read when the context is created); same frame, checksum and ray count must agree.
usage: tools/exp_queue_order.py [--spp N] [--size WxH] [--scene S]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %r)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
W, H, spp, sc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
r = srt.Renderer(0)
scene = srt.Scene.builtin(sc, 0).build_bvh(1, 1984)
r.upload_scene(scene); r.set_camera(scene.default_camera(W, H)); r.set_partition(0, 1)
ms = []
for _ in range(3):
    r.init_device_params(W, H, spp, 16, 1984)
    r.render_chunk(W, H); r.synchronize()
    ms.append(round(r.last_kernel_ms(), 1))
r.scatter_tiles()
cs = int(sum(int(p.astype("int64").sum()) for p in r.read_fb()))
print("RESULT " + json.dumps({"ms": ms, "checksum": cs, "rays": r.stats()["rays"]}))
''' % ROOT
def opt(name, d):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else d
spp = int(opt("--spp", "256")); W, H = (int(x) for x in opt("--size", "3840x2160").split("x")); sc = int(opt("--scene", "101"))
settings = [("default", {}), ("SRT_PROBE_SPP=0 (raster order)", {"SRT_PROBE_SPP": "0"}), ("SRT_ORDER_MAX_PCT=100", {"SRT_ORDER_MAX_PCT": "100"}),
            ("SRT_SPLIT_LOAD=0", {"SRT_SPLIT_LOAD": "0"}), ("default again", {})]
if "--more" in sys.argv:      # the finer sweep (cfg 2: 3.5 pixels per lane, the tail is 19.5 % of the wave-slot time)
    settings = [("default", {}), ("SRT_ORDER_MAX_PCT=50", {"SRT_ORDER_MAX_PCT": "50"}), ("SRT_ORDER_MAX_PCT=100", {"SRT_ORDER_MAX_PCT": "100"}),
                ("SRT_ORDER_MAX_PCT=150", {"SRT_ORDER_MAX_PCT": "150"}), ("SRT_PROBE_SPP=4", {"SRT_PROBE_SPP": "4"}), ("SRT_PROBE_SPP=8", {"SRT_PROBE_SPP": "8"}),
                ("SRT_PROBE_SPP=8 SRT_ORDER_MAX_PCT=100", {"SRT_PROBE_SPP": "8", "SRT_ORDER_MAX_PCT": "100"}), ("SRT_SPLIT_LOAD=0", {"SRT_SPLIT_LOAD": "0"}),
                ("SRT_SPLIT_LOAD=100", {"SRT_SPLIT_LOAD": "100"}), ("SRT_SPLIT_LOAD=400", {"SRT_SPLIT_LOAD": "400"}), ("default again", {})]
ref = None
for name, extra in settings:
    env = dict(os.environ); env.update(extra)
    p = subprocess.run([sys.executable, "-c", CHILD, str(W), str(H), str(spp), str(sc)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("%-34s FAILED: %s" % (name, p.stderr[-600:]), flush=True); continue
    o = json.loads(line[0][7:])
    key = (o["checksum"], o["rays"])
    if ref is None: ref = key
    print("%-34s scene %d %dx%d x %d spp: %s ms  (%.0f Mray/s)  %s" % (name, sc, W, H, spp, o["ms"], o["rays"] / min(o["ms"]) / 1e3, "exact" if key == ref else "DIFFERENT " + str(key)), flush=True)
