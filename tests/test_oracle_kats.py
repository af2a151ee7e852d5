"""Pins the CPU oracle against the only reference-produced values that exist for this path: the function-level
known answers recorded in SURVEY.md 8(c) (tests/golden/survey_kats.json).  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_kats.json")))
f32 = lambda s: np.float32(s)


def test_struct_sizes(orc):
    assert orc.lib().orc_sizeof(0) == G["struct_sizes"]["material"]
    assert orc.lib().orc_sizeof(1) == G["struct_sizes"]["camera_data"]
    assert C.sizeof(orc.Material) == 428 and C.sizeof(orc.CameraData) == 84


def test_xorwow(orc):
    L = orc.lib()
    s = orc.Rng()
    L.orc_rng_init(1984, C.byref(s))
    assert [L.orc_rng_next(C.byref(s)) for _ in range(4)] == G["xorwow_seed1984_first4_u32"]
    L.orc_rng_init(1984, C.byref(s))
    got = [np.float32(L.orc_random_float(C.byref(s))) for _ in range(4)]
    assert got == [f32(v) for v in G["xorwow_seed1984_first4_uniform"]]
    # uniform is in (0, 1]: x = 0xffffffff maps to 1.0, x = 0 to 2^-33
    assert np.float32(np.float32(0xffffffff) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10) / np.float32(2)) == np.float32(1.0)


def test_random_int_never_two(orc):
    L = orc.lib()
    s = orc.Rng()
    L.orc_rng_init(1984, C.byref(s))
    draws = [L.orc_random_int(0, 2, C.byref(s)) for _ in range(100000)]
    assert set(draws) == {0, 1}
    assert abs(draws.count(0) - 50000) < 1000


def test_hero_and_xyz(orc):
    L = orc.lib()
    wl = (C.c_float * 7)()
    L.orc_hero_wavelengths(1984, wl)
    assert [np.float32(v) for v in wl] == [f32(v) for v in G["hero_wavelengths_seed1984"]]
    pw = (C.c_float * 7)(*([1.0] * 7))
    out = (C.c_float * 3)()
    L.orc_spectrum_to_XYZ(wl, pw, 7, out)
    assert [np.float32(v) for v in out] == [f32(v) for v in G["spectrum_to_XYZ_unit_power_valid7"]]
    L.orc_spectrum_to_XYZ(wl, pw, 1, out)
    assert [np.float32(v) for v in out] == [f32(v) for v in G["spectrum_to_XYZ_unit_power_valid1"]]
    L.orc_spectrum_to_XYZ(wl, pw, 0, out)
    assert list(out) == [0.0, 0.0, 0.0]                          # Q7: valid = 0 contributes nothing


def test_xyz_to_srgb(orc):
    k = G["XYZ_to_sRGB"]
    lin, q = (C.c_float * 3)(), (C.c_float * 3)()
    orc.lib().orc_XYZ_to_sRGB((C.c_float * 3)(*k["xyz"]), lin, q)
    assert [np.float32(v) for v in lin] == [f32(v) for v in k["linear"]]
    assert [int(v) for v in q] == k["quantised"]
    # gamma branches (color.cu:15-22)
    cc = orc.lib().orc_correct_channel
    assert cc(-0.5) == 0.0 and cc(1.5) == 1.0 and cc(1.0) == 1.0
    assert np.float32(cc(0.001)) == np.float32(12.92) * np.float32(0.001)


def test_reflectance_refract(orc):
    L = orc.lib()
    k = G["reflectance"]
    ref_idx = np.float32(k["ref_idx_num"]) / np.float32(k["ref_idx_den"])
    assert np.float32(L.orc_reflectance(k["cosine"], ref_idx)) == f32(k["value"])
    k = G["refract"]
    uv = (C.c_float * 3)()
    L.orc_unit_vector((C.c_float * 3)(*k["uv_unnormalised"]), uv)
    out = (C.c_float * 3)()
    L.orc_refract(uv, (C.c_float * 3)(*k["n"]), np.float32(k["eta_num"]) / np.float32(k["eta_den"]), out)
    assert [np.float32(v) for v in out] == [f32(v) for v in k["value"]]


def test_spectrum_interp(orc):
    L = orc.lib()
    assert np.float32(L.orc_cie_interp(1, 555.0)) == f32(G["spectrum_interp"]["cie_y_555"])
    assert np.float32(L.orc_cie_interp(3, 560.0)) == f32(G["spectrum_interp"]["normalized_d65_560"])
    # table points are reproduced exactly, ends clamp (offset in [0, 93])
    for k in (0, 1, 40, 93):
        assert L.orc_cie_interp(0, 360.0 + 5.0 * k) == L.orc_cie_table(0, k)
    assert L.orc_cie_interp(1, 830.0) == L.orc_cie_table(1, 94)
    # CIE y-bar integrates (5 nm Riemann sum) to the reference's CIE_Y_INTEGRAL (cie_const.cuh:10)
    ysum = sum(L.orc_cie_table(1, k) for k in range(95)) * 5.0
    assert abs(ysum - 106.856895) < 1e-3


def test_baked_spectra(orc):
    L = orc.lib()
    m = orc.Material()
    m.col[:] = [1, 1, 1]; m.material_type = 4; m.emission_power = 5
    assert L.orc_material_bake(C.byref(m)) == 1
    k = G["light_1_1_1_power5_baked"]
    assert [np.float32(m.spectral_distribution[i]) for i in k["indices"]] == [f32(v) for v in k["values"]]
    m = orc.Material()
    m.col[:] = [.73, .73, .73]; m.material_type = 0
    assert L.orc_material_bake(C.byref(m)) == 1
    assert set(m.spectral_distribution) == {1.0}                 # Q2: albedo 0.73 saturates to 1.0
    m.col[:] = [.5, .5, .5]
    L.orc_material_bake(C.byref(m))
    assert set(m.spectral_distribution) == {0.5}
    m.col[:] = [.3, .3, .3]
    L.orc_material_bake(C.byref(m))
    assert set(m.spectral_distribution) == {0.0}
    m.col[:] = [.65, .05, .05]
    assert L.orc_material_bake(C.byref(m)) == 0                  # needs the rgb2spec table (absent upstream)
    bg = np.zeros(95, np.float32)
    assert L.orc_background_spectrum((C.c_float * 3)(0, 0, 0), orc.fptr(bg)) == 1 and not bg.any()   # black background


def test_sellmeier_quirk_q1(orc):
    L = orc.lib()
    B = (C.c_float * 3)(1.34533359, 0.209073176, 0.937357162)
    k = G["sellmeier_flint_C_equals_B"]
    for lam in ("360", "458", "480", "550", "600"):
        assert abs(L.orc_sellmeier_index(B, B, float(lam)) - k[lam]) < 2e-3
    for lam in k["nan_at"]:
        assert math.isnan(L.orc_sellmeier_index(B, B, float(lam)))


def test_camera_q23(orc):
    cam = orc.CameraData()
    orc.lib().orc_camera_init(256, 256, 40.0, orc.f3((278, 278, -800)), orc.f3((278, 278, 0)), orc.f3((0, 1, 0)), 0.0, 10.0, C.byref(cam))
    k = G["camera_cornell_256"]
    assert [np.float32(v) for v in cam.pixel00_loc] == [f32(v) for v in k["p00"]]
    assert [np.float32(v) for v in cam.pixel_delta_u] == [f32(v) for v in k["du"]]
    assert [np.float32(v) for v in cam.pixel_delta_v] == [f32(v) for v in k["dv"]]


def test_powf_spec_vs_libm(orc):
    """srt_powf (DESIGN D3) is exp(y log x) in fp64 rounded once: it must agree with correctly rounded pow
    everywhere except (at most) a vanishing number of near-tie inputs."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(40000), 10.0 ** rng.uniform(-30, 0, 20000)]).astype(np.float32)
    for y in (5.0, 0.416666, 2.0):
        got = np.array([orc.lib().orc_powf(float(v), float(np.float32(y))) for v in x], np.float32)
        want = np.power(x.astype(np.float64), np.float64(np.float32(y))).astype(np.float32)
        assert int(np.sum(got.view(np.uint32) != want.view(np.uint32))) <= 1
    p = orc.lib().orc_powf
    assert p(0.0, 5.0) == 0.0 and p(1.0, 5.0) == 1.0 and p(5.0, 2.0) == 25.0 and p(0.25, 0.0) == 1.0
    assert math.isnan(p(float("nan"), 5.0))
