#!/usr/bin/env python3
"""Freezes outputs of the CPU oracle: runs oracle/srt_oracle.c (no GPU) on the workloads of tests/helpers.py::digest_workloads and
writes the sha256 of the nine output planes + ray / path counts to tests/golden/oracle_digests.json.

Why: product and oracle restate the same deviations (D1 draw order, D3 srt_powf, no FMA) separately but are edited by the same hand; an
edit that moves both the same way keeps every GPU == oracle test green.  tests/test_oracle_digests.py re-derives these digests from the
oracle on the CPU and compares the HIP path with them on the GPU.  This is a drift guard, NOT parity evidence against the reference
(the reference's render path cannot be built here).  Regenerate only for a deliberate, documented change:
    python tests/golden/make_oracle_digests.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
import oracle_binding as O      # noqa: E402
from helpers import digest_of_render, digest_workloads, oracle_scene_for      # noqa: E402

doc = {"_about": "sha256 of the fp32 bit patterns of the nine block-linear output planes of the CPU oracle + ray / path counts; made by tests/golden/make_oracle_digests.py"}
for name, (scene, cam, W, H, spp, depth, mode) in digest_workloads(srt).items():
    ref = oracle_scene_for(O, scene, mode).render(cam, W, H, spp, depth)
    doc[name] = dict(digest_of_render(ref), width=W, height=H, spp=spp, depth=depth, tris=scene.n_tris, builder="reference" if mode == 0 else "SAH")
    print(name, doc[name]["rays"], doc[name]["xyz_x"][:16])
json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json"), "w"), indent=1, sort_keys=True)
