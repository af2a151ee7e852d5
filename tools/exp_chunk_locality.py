#!/usr/bin/env python3
"""Does cfg 5's kernel gain from L2 locality?  The 100k-triangle mesh's records (13.6 MB) do not fit an XCD's 4 MB L2 (hit rate 88 %), and a
traversal step waits for its slowest lane: with ~40 lanes per step nearly every step waits for an L2 miss.  If the CUs of an XCD worked on ONE
screen region, each L2 would hold that region's part of the tree.  Proxy without a kernel change: render the frame as nx x ny chunks one after the
other (every chunk uses the whole GPU, so all eight L2s hold the same region at a time) and compare the summed kernel time with the whole frame's.
usage (through gpurun): python tools/exp_chunk_locality.py [spp]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 3840, 2160
scene = srt.Scene.builtin(101, 0).build_bvh(1, 1984)
cam = scene.default_camera(W, H)
r = srt.Renderer(0)
r.upload_scene(scene); r.set_camera(cam); r.set_partition(0, 1)
for nx, ny in ((1, 1), (2, 1), (2, 2), (4, 2), (4, 4), (8, 4), (1, 1)):
    cw, ch = W // nx, H // ny
    total, rays = 0.0, 0
    for rep in range(2):
        total, rays = 0.0, 0
        r.init_device_params(cw, ch, spp, 16, 1984)
        for j in range(ny):
            for i in range(nx):
                r.render_chunk(cw, ch, i * cw, j * ch); r.synchronize()
                total += r.last_kernel_ms(); rays += r.stats()["rays"]
    print("%d x %d chunks of %dx%d at %d spp: kernel time %.1f ms in all, %d rays, %.0f Mray/s" % (nx, ny, cw, ch, spp, total, rays, rays / total / 1e3), flush=True)
