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
