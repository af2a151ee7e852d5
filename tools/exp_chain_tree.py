#!/usr/bin/env python3
"""A tree for chain-bound launches?  Child order profiled from the lanes of the most expensive tiles only (SRT_ORDER_PROFILE_CHAIN_SHARE,
accepted when the probe frame's most expensive pixel got cheaper), on the builder's tree, against cfg 2 and cfg 3's W = 8 / 4 shares.
usage: tools/exp_chain_tree.py"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
r = srt.Renderer(0)
def frames(scene, W, H, spp, rank, world, reps):
    r.upload_scene(scene); r.set_camera(scene.default_camera(W, H)); r.set_partition(rank, world)
    ms = []
    for _ in range(reps):
        r.init_device_params(W, H, spp, 16, 1984); r.render_chunk(W, H); r.synchronize(); ms.append(round(r.last_kernel_ms(), 1))
    return ms
for share in (0.0, -1.0, 0.25, 0.5, 0.75):
    scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
    note = "builder's tree"
    if share != 0.0:
        if share > 0:
            os.environ["SRT_ORDER_PROFILE_CHAIN_SHARE"] = str(share)
        else:
            os.environ.pop("SRT_ORDER_PROFILE_CHAIN_SHARE", None)
        pw, ph = 640, 360
        r.set_camera(scene.default_camera(pw, ph))
        n = r.order_children_by_profile(scene, pw, ph, 12, 16, 8)
        note = ("votes of tiles >= %.2f of the most expensive" % share if share > 0 else "all votes (the throughput rule)") + ", %d nodes swapped" % n
    out = {"tree": note, "cfg2_1280x720x256": frames(scene, 1280, 720, 256, 0, 1, 4)}
    for w, ranks in ((8, (7, 1, 3)), (4, (3,))):
        out["cfg3_W%d_ranks_%s" % (w, "_".join(map(str, ranks)))] = [min(frames(scene, 1920, 1080, 1024, k, w, 2)) for k in ranks]
    print(json.dumps(out), flush=True)
