#!/usr/bin/env python3
"""Census for the "quad" visit: how often would it fire, and with what outcome?  Host only (no GPU).

A quad parent P is an INNER node (both children internal) whose children L and R both have two leaf children.  A quad record would
replace P's INNER visit (two box tests, LDS, 59 instructions in the hand-scheduled burst) and the FRINGE visits of L and R by ONE visit
that tests both boxes and up to four triangles in the reference's order (bvh/bvh.cu:120-162).  This script walks the headline scene's tree
(the throughput-tuned one bench.py renders) for camera rays and for secondary rays started at their hit points with cosine-free random
directions, in the reference's order with the reference's pruning (closest_so_far), in float64 (a census, not a parity tool), and counts

  * INNER / FRINGE visits per ray, and how many of them sit at quad parents / under quad parents,
  * for the visits at quad parents: neither box passes / left only / right only / both.

usage: python tools/quad_census.py [n_rays]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sc = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
sc.optimise_bvh(3)
left, right, prim, boxes = sc.bvh()
tris = sc.triangles()
V = np.array([[list(t.v0), list(t.v1), list(t.v2)] for t in tris], np.float64)
n = len(left)
leaf = prim >= 0
kind = np.zeros(n, int)      # 0 leaf, 1 LL, 2 LI, 3 II
for k in range(n):
    if not leaf[k]:
        a, b = leaf[left[k]], leaf[right[k]]
        kind[k] = 1 if (a and b) else (2 if (a or b) else 3)
quad = np.array([kind[k] == 3 and kind[left[k]] == 1 and kind[right[k]] == 1 for k in range(n)])
under_quad = np.zeros(n, bool)
for k in np.nonzero(quad)[0]:
    under_quad[left[k]] = under_quad[right[k]] = True
print("tree: %d triangles, %d INNER (II) nodes of which %d quad parents, %d LL nodes of which %d under a quad parent, %d LI nodes" %
      (leaf.sum(), (kind == 3).sum(), quad.sum(), (kind == 1).sum(), under_quad.sum(), (kind == 2).sum()))

def box_hit(k, o, inv, c):
    b = boxes[k]
    t0 = (b[0::2] - o) * inv; t1 = (b[1::2] - o) * inv
    lo = np.minimum(t0, t1); hi = np.maximum(t0, t1)
    return not (min(c, hi.min()) <= max(0.0, lo.max()))

def tri_hit(t, o, d, c):
    v0, v1, v2 = V[t]
    nrm = np.cross(v1 - v0, v2 - v0); ln = np.linalg.norm(nrm)
    if ln == 0: return None
    nrm /= ln
    den = nrm @ d
    if abs(den) < 1e-8: return None
    tt = (nrm @ v0 - nrm @ o) / den
    if not (0.0 <= tt <= c): return None
    p = o + tt * d
    s = [np.dot(np.cross(b - a, p - a), nrm) for a, b in ((v0, v1), (v1, v2), (v2, v0))]
    return tt if (all(x >= 0 for x in s) or all(x <= 0 for x in s)) else None

stat = dict(rays=0, inner=0, fringe=0, fringe_li=0, inner_at_quad=0, fringe_under_quad=0, q_none=0, q_left=0, q_right=0, q_both=0, tri=0)
def trace(o, d):
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    c, hit = 3.4e38, -1
    stack, node = [], 0
    stat["rays"] += 1
    while True:
        l, r = left[node], right[node]
        res = []
        if kind[node] == 3: stat["inner"] += 1
        else: stat["fringe"] += 1; stat["fringe_under_quad"] += int(under_quad[node]); stat["fringe_li"] += int(kind[node] == 2)
        for ch in (l, r):
            if leaf[ch]:
                stat["tri"] += 1
                t = tri_hit(prim[ch], o, d, c)
                if t is not None: c, hit = t, prim[ch]
                res.append(False)
            else:
                res.append(box_hit(ch, o, inv, c))
        if quad[node]:
            stat["inner_at_quad"] += 1
            stat["q_both" if all(res) else "q_left" if res[0] else "q_right" if res[1] else "q_none"] += 1
        if res[0]:
            if res[1]: stack.append(r)
            node = l
        elif res[1]: node = r
        elif stack: node = stack.pop()
        else: break
    return c, hit

cam = sc.default_camera(1920, 1080)
rng = np.random.default_rng(7)
p00, du, dv, ctr = (np.array(list(x), np.float64) for x in (cam.pixel00_loc, cam.pixel_delta_u, cam.pixel_delta_v, cam.camera_center))
for phase in ("camera rays", "secondary rays (from the hit points, uniform random directions)"):
    for k in stat: stat[k] = 0
    hits = []
    if phase.startswith("camera"):
        for _ in range(n_rays):
            i, j = rng.uniform(0, 1920), rng.uniform(0, 1080)
            d = p00 + i * du + j * dv - ctr
            c, h = trace(ctr, d)
            if h >= 0: hits.append(ctr + c * d)
        first_hits = hits
    else:
        for p in first_hits:
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            trace(p + 1e-4 * d, d)
    r = max(stat["rays"], 1)
    print("%s: %d rays: INNER visits %.2f / ray (%.2f at quad parents), FRINGE visits %.2f / ray (%.2f under quad parents, %.2f at leaf + subtree nodes), triangle tests %.2f / ray" %
          (phase, r, stat["inner"] / r, stat["inner_at_quad"] / r, stat["fringe"] / r, stat["fringe_under_quad"] / r, stat["fringe_li"] / r, stat["tri"] / r))
    q = max(stat["inner_at_quad"], 1)
    print("   visits at quad parents by outcome: neither box %.1f %%, left only %.1f %%, right only %.1f %%, both %.1f %%" %
          (100 * stat["q_none"] / q, 100 * stat["q_left"] / q, 100 * stat["q_right"] / q, 100 * stat["q_both"] / q))
