#!/usr/bin/env python3
"""Tree quality experiment: the SAH builder's insertion-based post-optimisation (SRT_BVH_OPT_PASSES) against node records / triangle
tests per ray (instrumented kernel) and render-kernel time (production kernel) on cfg 3's and cfg 5's scenes.
usage: tools/exp_bvh_opt.py [passes ...]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
passes = [int(x) for x in sys.argv[1:]] or [0, 1, 3, 6]
only = [int(x) for x in os.environ.get("EXP_SCENES", "100,101").split(",")]
r = srt.Renderer(0)
for sid, W, H, spp, reps in ((100, 1920, 1080, 256, 3), (101, 1920, 1080, 128, 2)):
    if sid not in only:
        continue
    for p in passes:
        os.environ["SRT_BVH_OPT_PASSES"] = str(p)
        t0 = time.time()
        scene = srt.Scene.builtin(sid, 0).build_bvh(1, 1984)
        build_s = time.time() - t0
        cam = scene.default_camera(W, H)
        r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
        plan = r.launch_plan()
        r.set_count_traversal(True)
        r.init_device_params(W, H, 16, 16, 1984); r.render_chunk(W, H); r.synchronize()
        st = r.stats()
        r.set_count_traversal(False)
        ms = []
        for _ in range(reps):
            r.init_device_params(W, H, spp, 16, 1984); r.render_chunk(W, H); r.synchronize(); ms.append(round(r.last_kernel_ms(), 2))
        r.scatter_tiles()
        cs = int(sum(int(q.astype("int64").sum()) for q in r.read_fb()))
        print(json.dumps({"scene": sid, "passes": p, "order": os.environ.get("SRT_BVH_ORDER", "0"), "build_s": round(build_s, 2), "depth": scene.bvh_depth, "plan": plan,
                          "V": round(st["node_visits"] / st["rays"], 3), "T": round(st["tri_tests"] / st["rays"], 3),
                          "kernel_ms": ms, "Gray_s": round(r.stats()["rays"] / min(ms) / 1e6, 3), "fb_checksum": cs}), flush=True)
