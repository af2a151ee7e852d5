"""Pinning of the constant tables against the REFERENCE ITSELF: utils/cie_const.cu and utils/color_const.cu compile unmodified
with this image's hipcc (-x hip; recipe: oracle/Makefile, target `ref`, output oracle/_ref/libref_tables.so, git-ignored, built
by __graft_entry__.build() where /root/reference is mounted).  The CIE 1931 colour matching functions, the normalised D65
illuminant and the XYZ -> sRGB matrix that the product (srt_color_tables) and the oracle compute with must equal the reference's
arrays bit for bit.  The other pieces of the reference that compile here without stand-ins for CUDA headers: sellmeier_index (below),
and the host files transform.cu / params.cpp / log_context.cpp / image.cpp / save_image.cpp (tests/test_ref_host.py); everything on
the render path proper needs curand_kernel.h and cannot be built."""
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


@pytest.mark.gpu
def test_reference_sellmeier_device_function(srt, orc, gpu):
    """The reference's only curand-free device function, sellmeier_index (refraction/sellmeier.cu:11-22), compiled UNMODIFIED from
    its source (oracle/Makefile `ref`: relocatable device object + a harness kernel of ours that calls it) and run on the GPU, against
    the product's device arithmetic (op-sweep kind 22) and the oracle's CPU arithmetic, on the reference's coefficient tables with
    and without quirk Q1 (C := B), on random coefficients, over 300..900 nm, the poles, zero, infinities and NaN.

    Two builds of the reference function.  -ffp-contract=off (the IEEE reading of the source this repository commits to): product and
    oracle must equal it BIT FOR BIT.  The compiler's default contraction (hipcc: fast -- nvcc's default -fmad=true is the same
    licence): `lambda * lambda - c` may become one fused multiply-add; that build is what the shipped CUDA binary most likely
    computes, and it is NOT bit-identical to the no-contraction reading -- the test measures by how much (the concrete face of the
    'no FMA contraction' deviation of DESIGN.md section 2) and only requires the two to stay close."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_sellmeier.so")
    so_c = os.path.join(ROOT, "oracle", "_ref", "libref_sellmeier_contract.so")
    if not (os.path.exists(so) and os.path.exists(so_c)):
        pytest.skip("oracle/_ref/libref_sellmeier*.so not built (the reference is not mounted here)")
    fp = C.POINTER(C.c_float)
    L, Lc = C.CDLL(so), C.CDLL(so_c)
    for lib in (L, Lc):
        lib.ref_sellmeier_run.argtypes = [fp, fp, fp, C.c_uint, fp]
    rng = np.random.default_rng(5)
    flint_b, flint_c = (1.34533359, 0.209073176, 0.937357162), (0.00997743871, 0.0470450767, 111.886764)      # refraction/sellmeier.cuh:14-15
    bk7_b, bk7_c = (1.03961212, 0.231792344, 1.01046945), (6.00069867e-3, 2.00179144e-2, 1.03560653e2)         # :6-7
    sets = [(flint_b, flint_b), (flint_b, flint_c), (bk7_b, bk7_b), (bk7_b, bk7_c)] + \
           [(tuple(rng.uniform(0.1, 2.0, 3)), tuple(rng.uniform(0.001, 120.0, 3))) for _ in range(4)]
    n = 1 << 14
    n_nan, n_total, n_contract_differs, ulps, n_nan_flips = 0, 0, 0, [], 0
    for b3, c3 in sets:
        b = np.array(b3, np.float32); c = np.array(c3, np.float32)
        lam = rng.uniform(300.0, 900.0, n).astype(np.float32)
        poles = np.sqrt(c.astype(np.float64)) * 1000.0            # lambda^2 (um^2) == C: division by zero
        special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, 1e30, 360.0, 830.0] + list(poles) + list(np.nextafter(poles.astype(np.float32), np.float32(0))), np.float32)
        lam[: special.size] = special
        ref, ref_c = np.zeros(n, np.float32), np.zeros(n, np.float32)
        assert L.ref_sellmeier_run(b.ctypes.data_as(fp), c.ctypes.data_as(fp), lam.ctypes.data_as(fp), n, ref.ctypes.data_as(fp)) == 0
        assert Lc.ref_sellmeier_run(b.ctypes.data_as(fp), c.ctypes.data_as(fp), lam.ctypes.data_as(fp), n, ref_c.ctypes.data_as(fp)) == 0
        coeffs = np.zeros(n, np.float32); coeffs[:3] = b; coeffs[3:6] = c
        got = gpu.op_sweep(22, lam, coeffs)
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert np.all(same), ("product vs reference device function (no contraction)", b3, c3, int(np.sum(~same)), lam[~same][:4], got[~same][:4], ref[~same][:4])
        want = np.array([orc.lib().orc_sellmeier_index(b.ctypes.data_as(fp), c.ctypes.data_as(fp), float(v)) for v in lam[:2048]], np.float32)
        same = (want.view(np.uint32) == ref[:2048].view(np.uint32)) | (np.isnan(want) & np.isnan(ref[:2048]))
        assert np.all(same), ("oracle vs reference device function (no contraction)", b3, c3, int(np.sum(~same)))
        n_nan += int(np.sum(np.isnan(ref))); n_total += n
        fin = np.isfinite(ref) & np.isfinite(ref_c)
        d = np.abs(ref.view(np.int32).astype(np.int64) - ref_c.view(np.int32).astype(np.int64))[fin]
        n_contract_differs += int(np.sum(d != 0)); ulps.append(d)
        n_nan_flips += int(np.sum(np.isnan(ref) != np.isnan(ref_c)))  # next to a pole the contracted build may land on the other side of it
    assert n_nan > 0                                   # quirk Q1 really produces NaN indices somewhere in the sweep
    assert n_nan_flips <= n_total // 1000              # ... and contraction moves an operand across NaN / finite only at a pole: a counted handful
    ulps = np.concatenate(ulps)
    frac = n_contract_differs / n_total
    print("contraction build differs from the IEEE reading in %.2f %% of the operands; median / 99th percentile / max ulp distance of those: %d / %d / %d"
          % (100 * frac, int(np.median(ulps[ulps > 0])) if frac else 0, int(np.percentile(ulps[ulps > 0], 99)) if frac else 0, int(ulps.max())))
    # what the FMA changes stays a rounding effect away from the poles: few operands, an ulp or two for nearly all of them
    assert frac < 0.05 and (frac == 0 or np.median(ulps[ulps > 0]) <= 2)
