#!/usr/bin/env python3
"""Renders a scene on the GPU and writes an 8-bit PNG (tiny zlib writer, no imaging library needed): eyeball check."""
import argparse, importlib, os, struct, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=100); ap.add_argument("--bvh", type=int, default=1)
ap.add_argument("--width", type=int, default=960); ap.add_argument("--height", type=int, default=540)
ap.add_argument("--spp", type=int, default=64); ap.add_argument("--depth", type=int, default=16)
ap.add_argument("--out", default="gpurun_out/render.png")
a = ap.parse_args()
scene = srt.Scene.builtin(a.scene, 0).build_bvh(a.bvh, 1984)
cam = scene.default_camera(a.width, a.height)
img = srt.render_image(scene, cam, a.width, a.height, a.spp, a.depth)
rgb = np.stack([p.reshape(a.height, a.width) for p in img["rowmajor"]], -1).astype(np.uint8)
raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(a.height))
def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", a.width, a.height, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
open(a.out, "wb").write(png)
print("wrote", a.out, rgb.shape, "mean", rgb.mean(axis=(0, 1)), "kernel ms", img["kernel_ms"])
