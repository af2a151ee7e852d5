#!/usr/bin/env python3
"""Child order from a ray profile (srt_order_children_by_profile) against the builder's nearer-child-first rule: node records /
triangle tests per ray (instrumented kernel) and render-kernel time (production kernel), cfg 3's and cfg 5's scenes.
usage: tools/exp_profile_order.py   (env SRT_ORDER_PROFILE_RULE, EXP_SCENES, EXP_PROBE="WxHxSPP,..." as wished)"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
only = [int(x) for x in os.environ.get("EXP_SCENES", "100,101").split(",")]
probes = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("EXP_PROBE", "0x0x0,480x270x8,960x540x16").split(",")]
r = srt.Renderer(0)
for sid, W, H, spp, reps in ((100, 1920, 1080, 256, 3), (101, 1920, 1080, 128, 2)):
    if sid not in only:
        continue
    for pw, ph, ps in probes:
        scene = srt.Scene.builtin(sid, 0).build_bvh(1, 1984)
        cam = scene.default_camera(W, H)
        swapped = None
        r.set_camera(scene.default_camera(pw, ph) if pw else cam)
        if pw:
            swapped = r.order_children_by_profile(scene, pw, ph, ps, 16, int(os.environ.get("EXP_MIN_SAMPLES", "4")))
        r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
        r.set_count_traversal(True)
        r.init_device_params(W, H, 16, 16, 1984); r.render_chunk(W, H); r.synchronize()
        st = r.stats()
        r.set_count_traversal(False)
        ms = []
        for _ in range(reps):
            r.init_device_params(W, H, spp, 16, 1984); r.render_chunk(W, H); r.synchronize(); ms.append(round(r.last_kernel_ms(), 2))
        r.scatter_tiles()
        cs = int(sum(int(q.astype("int64").sum()) for q in r.read_fb()))
        print(json.dumps({"scene": sid, "probe": "%dx%dx%d" % (pw, ph, ps) if pw else "none (builder's order)", "rule": os.environ.get("SRT_ORDER_PROFILE_RULE", "0"),
                          "nodes_swapped": swapped, "V": round(st["node_visits"] / st["rays"], 3), "T": round(st["tri_tests"] / st["rays"], 3),
                          "kernel_ms": ms, "Gray_s": round(r.stats()["rays"] / min(ms) / 1e6, 3), "fb_checksum": cs}), flush=True)
