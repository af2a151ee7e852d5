#!/usr/bin/env python3
"""How far does FMA contraction move an image?  Renders the same frame with the shipped library (no contraction: the IEEE reading of the
reference's source, bit-identical to the oracle) and with a variant of the kernels built with the compiler's default contraction
(tools/build_variant.sh contract -ffp-contract=fast -- the licence nvcc's default -fmad=true gives the real reference binary), in two
child processes, and compares the unquantised sRGB planes.  One flipped discrete decision (an edge hit, a Fresnel branch, a rejection
accept) shifts the pixel's RNG stream, so the pixel becomes a different Monte-Carlo estimate of the same integral: the question is how
many pixels that happens to and how large the difference is -- the context of BASELINE's 1e-3 L-inf tolerance.
usage (GPU box): python tools/contraction_effect.py [--spp 64 1024]"""
import argparse, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
srt = importlib.import_module('cuda-spectral-ray-tracer_amd')
scene = srt.Scene.builtin(%d, 0).build_bvh(%d, 1984)
W, H, spp = %d, %d, %d
out = srt.render_image(scene, scene.default_camera(W, H), W, H, spp, 16)
np.save(%r, np.stack(list(out['lin']) + list(out['fb'])))
"""
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, nargs="+", default=[64, 1024])
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
variant = os.path.join(ROOT, "gpurun_exp_contract.so")
assert os.path.exists(variant), "build it first: tools/build_variant.sh contract -ffp-contract=fast"
for spp in a.spp:
    imgs = []
    for lib in (None, variant):
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, "img.npy")
            env = dict(os.environ)
            if lib:
                env["SRT_LIB_PATH"] = lib
            subprocess.check_call([sys.executable, "-c", CHILD % (ROOT, a.scene, a.bvh, a.width, a.height, spp, f)], env=env)
            imgs.append(np.load(f))
    lin0, lin1, q0, q1 = imgs[0][:3], imgs[1][:3], imgs[0][3:], imgs[1][3:]
    live = np.any(lin0 != 0, axis=0) | np.any(lin1 != 0, axis=0)          # lanes of the block-linear planes that are pixels
    d = np.abs(lin0 - lin1)[:, live]
    pix = np.any(lin0 != lin1, axis=0)[live]
    print(json.dumps({"spp": spp, "pixels": int(live.sum()), "pixels_with_any_different_bit": int(pix.sum()), "fraction": float(pix.mean()),
                      "linf_srgb": float(d.max()), "pixels_over_1e-3": int(np.sum(np.any(d > 1e-3, axis=0))), "fraction_over_1e-3": float(np.mean(np.any(d > 1e-3, axis=0))),
                      "mean_abs_diff": float(d.mean()), "quantised_values_different": int(np.sum((q0 != q1)[:, live])),
                      "mean_image_difference_of_channel_means": [float(abs(lin0[c][live].mean() - lin1[c][live].mean())) for c in range(3)]}))
