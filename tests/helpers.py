import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_planes_equal(got, want, what):
    for c, (a, b) in enumerate(zip(got, want)):
        ba, bb = bits(a), bits(b)
        if not np.array_equal(ba, bb):
            bad = np.nonzero(ba != bb)[0]
            raise AssertionError("%s plane %d: %d of %d lanes differ, first idx %d: got %r want %r" %
                                 (what, c, bad.size, ba.size, bad[0], a[bad[0]], b[bad[0]]))


def oracle_scene_for(O, scene, mode, seed=1984):
    """Oracle scene fed with the product's raw inputs; reference topology is rebuilt by the oracle itself,
    any other tree is imported (boxes are always recomputed by the oracle)."""
    osc = O.OracleScene(scene.triangles(), scene.materials(), scene.background())
    if mode == 0:
        assert osc.build_reference(seed) == 1
    else:
        left, right, prim, _ = scene.bvh()
        assert osc.set_bvh(left, right, prim, 0) == 1
    return osc


def custom_scene(srt, tris, mats, bg_rgb=(0.5, 0.5, 0.5)):
    """Scene from raw arrays (the boundary's srt_scene_set_* path): tris = [(v0, v1, v2, mat, aa_plane)], mats = [(type, rgb, fuzz, power)]."""
    import ctypes as C
    B = srt.binding
    T = (B.TriIn * len(tris))()
    for k, (v0, v1, v2, mat, aap) in enumerate(tris):
        T[k].v0[:] = v0; T[k].v1[:] = v1; T[k].v2[:] = v2; T[k].mat_index = mat; T[k].aa_plane = aap
    M = (B.Material * len(mats))()
    for k, (mtype, rgb, fuzz, power) in enumerate(mats):
        M[k].col[:] = rgb; M[k].reflection_fuzz = fuzz; M[k].material_type = mtype; M[k].emission_power = power
        M[k].sellmeier_B[:] = (1.03961212, 0.231792344, 1.01046945); M[k].sellmeier_C[:] = (1.03961212, 0.231792344, 1.01046945)   # Q1: C := B
        B.check(B.lib().srt_material_bake(C.byref(M[k])))
    bg = np.zeros(B.N_CIE, np.float32)
    B.check(B.lib().srt_background_spectrum((C.c_float * 3)(*bg_rgb), B.fptr(bg)))
    return srt.Scene.from_arrays(T, M, bg)


def fuzz_case(srt, seed):
    """Random-scene case `seed` of the fuzz tests: a triangle soup with random materials, camera and builder (both BVH builders, lens
    on / off, thin and axis-aligned triangles, shared edges and vertices so that exact t ties occur -- Q11).  Host side only (no GPU).
    Returns scene (BVH built), camera, W, H, spp, depth, builder mode, number of triangles."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 260))
    tris, mats = [], []
    n_mats = int(rng.integers(1, 12))
    for k in range(n_mats):
        mtype = int(rng.choice([0, 0, 0, 1, 1, 2, 4, 6]))
        grey = float(rng.choice([0.0, 0.3, 0.5, 0.73, 1.0]))
        mats.append((mtype, (grey, grey, grey), float(rng.uniform(0, 0.6)), float(rng.uniform(0.5, 3.0))))
    verts = rng.uniform(-5, 5, (max(4, n // 2), 3)).astype(np.float32)
    lattice = rng.random(verts.shape[0]) < 0.3
    verts[lattice] = np.round(verts[lattice])          # some vertices on lattice points: coplanar / axis-aligned coincidences
    for k in range(n):
        if rng.random() < 0.6:          # triangles that share vertices (a mesh-like soup: shared edges)
            i0, i1, i2 = rng.choice(verts.shape[0], 3, replace=False)
            v0, v1, v2 = verts[i0], verts[i1], verts[i2]
        else:
            c = rng.uniform(-5, 5, 3)
            v0, v1, v2 = (c + rng.normal(0, rng.choice([0.01, 0.5, 2.0]), 3) for _ in range(3))
        if rng.random() < 0.15:         # axis-aligned: exercises the aa_plane projection choice (tri.cu:66-77)
            ax = int(rng.integers(0, 3)); v0 = np.array(v0); v1 = np.array(v1); v2 = np.array(v2)
            v1[ax] = v0[ax]; v2[ax] = v0[ax]
        tris.append((tuple(float(x) for x in v0), tuple(float(x) for x in v1), tuple(float(x) for x in v2), int(rng.integers(0, n_mats)), int(rng.choice([0, 0, 1, 2, 3]))))
    bg = float(rng.choice([0.5, 1.0, 0.5, 0.0]))
    mode = int(rng.integers(0, 2))
    scene = custom_scene(srt, tris, mats, (bg, bg, bg)).build_bvh(mode, 1984)
    W, H, spp, depth = int(rng.integers(9, 70)), int(rng.integers(9, 50)), int(rng.integers(1, 7)), int(rng.integers(1, 17))
    cam = srt.camera_init(W, H, float(rng.uniform(20, 90)), tuple(rng.uniform(-12, 12, 3)), tuple(rng.uniform(-2, 2, 3)),
                          defocus_angle=float(rng.choice([0.0, 0.0, 1.5])), focus_dist=float(rng.uniform(5, 15)))
    return scene, cam, W, H, spp, depth, mode, n


# ---- frozen outputs of the CPU oracle (tests/golden/oracle_digests.json) ---------------------------------------------------
DIGEST_PLANES = ("fb_r", "fb_g", "fb_b", "srgb_r", "srgb_g", "srgb_b", "xyz_x", "xyz_y", "xyz_z")


def digest_workloads(srt):
    """The frozen workloads: PRISM 64x64 x 16 spp depth 8, BASELINE cfg 1 (CORNELL 256x256 x 16 spp depth 8; the coloured walls use
    this build's own sigmoid fit, as every CORNELL render here does), six fuzz seeds.  name -> (scene, cam, W, H, spp, depth, mode)."""
    out = {}
    sc = srt.Scene.builtin(srt.SCENE_PRISM).build_bvh(srt.BVH_REFERENCE, 1984)
    out["prism_64x64_16spp_d8"] = (sc, sc.default_camera(64, 64), 64, 64, 16, 8, 0)
    sc = srt.Scene.builtin(srt.SCENE_CORNELL).build_bvh(srt.BVH_REFERENCE, 1984)
    out["cfg1_cornell_256x256_16spp_d8"] = (sc, sc.default_camera(256, 256), 256, 256, 16, 8, 0)
    for seed in (0, 1, 2, 3, 4, 5):
        scene, cam, W, H, spp, depth, mode, _ = fuzz_case(srt, seed)
        out["fuzz_seed_%d" % seed] = (scene, cam, W, H, spp, depth, mode)
    return out


def digest_of_render(res):
    """sha256 of the bit patterns of the nine planes (quantised framebuffer, unquantised sRGB, XYZ sums: rendering/rendering.cu:205-234)
    + the ray / path counts of a render result (oracle or HIP path: same dict layout)"""
    import hashlib
    planes = list(res["fb"]) + list(res["lin"]) + list(res["xyz"])
    d = {name: hashlib.sha256(bits(p).tobytes()).hexdigest() for name, p in zip(DIGEST_PLANES, planes)}
    d["rays"] = int(res["stats"]["rays"]); d["paths"] = int(res["stats"]["paths"])
    return d
