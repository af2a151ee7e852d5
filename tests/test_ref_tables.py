"""Pinning of the constant tables against the REFERENCE ITSELF: utils/cie_const.cu and utils/color_const.cu compile unmodified
with this image's hipcc (-x hip; recipe: oracle/Makefile, target `ref`, output oracle/_ref/libref_tables.so, git-ignored, built
by __graft_entry__.build() where /root/reference is mounted).  The CIE 1931 colour matching functions, the normalised D65
illuminant and the XYZ -> sRGB matrix that the product (srt_color_tables) and the oracle compute with must equal the reference's
arrays bit for bit.  Nothing else of the reference compiles here without stand-ins for CUDA headers, so this is the whole extent
of "the reference compiled here" (DESIGN.md section 2)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_tables.so")
N = 95


def _ref():
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/utils"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_tables.so not built (the reference is not mounted here)")
    return C.CDLL(REF_SO)


def _array(L, name, n):
    return np.ctypeslib.as_array((C.c_float * n).in_dll(L, name)).copy()


def test_reference_tables_equal_product_and_oracle(srt, orc):
    L = _ref()
    ref = {k: _array(L, k, N) for k in ("cie_x", "cie_y", "cie_z", "normalized_cie_d65", "cie_d65")}
    ref_m = _array(L, "d65_XYZ_to_sRGB", 9)
    # product
    cmf, m = np.zeros(4 * N, np.float32), np.zeros(9, np.float32)
    fp = C.POINTER(C.c_float)
    assert srt.binding.lib().srt_color_tables(cmf.ctypes.data_as(fp), m.ctypes.data_as(fp)) == 0
    cmf = cmf.reshape(N, 4)
    for col, name in enumerate(("cie_x", "cie_y", "cie_z", "normalized_cie_d65")):
        assert np.array_equal(cmf[:, col].view(np.uint32), ref[name].view(np.uint32)), "product " + name
        want = np.array([orc.lib().orc_cie_table(col, k) for k in range(N)], np.float32)
        assert np.array_equal(want.view(np.uint32), ref[name].view(np.uint32)), "oracle " + name
    assert np.array_equal(m.view(np.uint32), ref_m.view(np.uint32)), "product XYZ->sRGB"
    om = np.zeros(9, np.float32)
    orc.lib().orc_color_matrix(om.ctypes.data_as(fp))
    assert np.array_equal(om.view(np.uint32), ref_m.view(np.uint32)), "oracle XYZ->sRGB"
    # sanity of what was loaded: y_bar peaks at 555 nm; D65 is normalised so that its y_bar-weighted integral (5 nm steps) is 1
    # (utils/cie_const.cu:83)
    assert int(np.argmax(ref["cie_y"])) == (555 - 360) // 5 and abs(float(ref["cie_y"].max()) - 1.0) < 1e-6
    assert abs(5.0 * float(np.sum(ref["normalized_cie_d65"].astype(np.float64) * ref["cie_y"].astype(np.float64))) - 1.0) < 1e-3
