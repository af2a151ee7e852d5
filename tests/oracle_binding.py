"""ctypes binding of the CPU oracle (oracle/srt_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("SRT_ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "libsrt_oracle.so")   # (override: sanitizer build)


class Rng(C.Structure):
    _fields_ = [("d", C.c_uint32), ("v", C.c_uint32 * 5)]


class Material(C.Structure):
    _fields_ = [("col", C.c_float * 3), ("reflection_fuzz", C.c_float), ("material_type", C.c_uint32),
                ("spectral_distribution", C.c_float * 95), ("emission_power", C.c_float),
                ("sellmeier_B", C.c_float * 3), ("sellmeier_C", C.c_float * 3)]


class CameraData(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3), ("pixel00_loc", C.c_float * 3),
                ("defocus_angle", C.c_float),
                ("camera_center", C.c_float * 3), ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3)]


class TriIn(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3),
                ("mat_index", C.c_uint32), ("aa_plane", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("trav_iters", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("max_stack", C.c_uint64)]


_fp = C.POINTER(C.c_float)
_lib = None


def build():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.orc_rng_init.argtypes = [C.c_uint64, C.POINTER(Rng)]
        L.orc_rng_next.restype = C.c_uint32
        L.orc_rng_next.argtypes = [C.POINTER(Rng)]
        L.orc_random_float.restype = C.c_float
        L.orc_random_float.argtypes = [C.POINTER(Rng)]
        L.orc_random_int.argtypes = [C.c_int, C.c_int, C.POINTER(Rng)]
        L.orc_powf.restype = C.c_float
        L.orc_powf.argtypes = [C.c_float, C.c_float]
        L.orc_spectrum_interp.restype = C.c_float
        L.orc_spectrum_interp.argtypes = [_fp, C.c_float, C.c_int]
        L.orc_correct_channel.restype = C.c_float
        L.orc_correct_channel.argtypes = [C.c_float]
        L.orc_sellmeier_index.restype = C.c_float
        L.orc_sellmeier_index.argtypes = [_fp, _fp, C.c_float]
        L.orc_reflectance.restype = C.c_float
        L.orc_reflectance.argtypes = [C.c_float, C.c_float]
        L.orc_cie_table.restype = C.c_float
        L.orc_cie_table.argtypes = [C.c_int, C.c_int]
        L.orc_cie_interp.restype = C.c_float
        L.orc_cie_interp.argtypes = [C.c_int, C.c_float]
        L.orc_color_matrix.argtypes = [_fp]
        L.orc_hero_wavelengths.argtypes = [C.c_uint64, _fp]
        L.orc_spectrum_to_XYZ.argtypes = [_fp, _fp, C.c_uint32, _fp]
        L.orc_XYZ_to_sRGB.argtypes = [_fp, _fp, _fp]
        L.orc_refract.argtypes = [_fp, _fp, C.c_float, _fp]
        L.orc_reflect.argtypes = [_fp, _fp, _fp]
        L.orc_unit_vector.argtypes = [_fp, _fp]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _fp]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_build_bvh_reference.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_scene_set_bvh.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_scene_node_count.restype = C.c_size_t
        L.orc_scene_node_count.argtypes = [C.c_void_p]
        L.orc_scene_get_bvh.restype = C.c_size_t
        L.orc_scene_get_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _fp]
        L.orc_scene_get_tris.argtypes = [C.c_void_p, _fp]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraData)] + [C.c_uint32] * 10 + [C.c_uint64, C.c_int, C.c_void_p,
                                 C.c_uint32, C.c_uint32] + [_fp] * 9 + [C.c_int, C.POINTER(Stats)]
        L.orc_unswizzle.argtypes = [_fp, _fp] + [C.c_uint32] * 9
        L.orc_camera_init.argtypes = [C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, C.c_float, C.c_float, C.POINTER(CameraData)]
        L.orc_bake_sigmoid_spectrum.argtypes = [_fp, C.c_float, C.c_int, _fp]
        L.orc_material_bake.argtypes = [C.POINTER(Material)]
        L.orc_background_spectrum.argtypes = [_fp, _fp]
        L.orc_trace_ray.argtypes = [C.c_void_p, _fp, _fp, _fp]
        L.orc_scatter.argtypes = [C.POINTER(Material), _fp, _fp, _fp, _fp, C.POINTER(C.c_uint32), _fp, _fp, C.c_float, C.c_int,
                                  C.c_uint64, C.POINTER(C.c_uint32)]
        L.orc_sizeof.restype = C.c_size_t
        L.orc_sizeof.argtypes = [C.c_int]
        _lib = L
    return _lib


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def fptr(a):
    return a.ctypes.data_as(_fp)


class OracleScene:
    """Oracle scene from the product's raw inputs (bytes-compatible structs)."""

    def __init__(self, tris, mats, background):
        self.n_tris, self.n_mats = len(tris), len(mats)
        bg = np.ascontiguousarray(background, np.float32)
        self._keep = (tris, mats, bg)
        self.h = C.c_void_p(lib().orc_scene_create(C.cast(tris, C.c_void_p), len(tris), C.cast(mats, C.c_void_p), len(mats), fptr(bg)))

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_reference(self, seed=1984):
        return lib().orc_scene_build_bvh_reference(self.h, seed)

    def set_bvh(self, left, right, prim, root=0):
        left, right, prim = (np.ascontiguousarray(a, np.int32) for a in (left, right, prim))
        return lib().orc_scene_set_bvh(self.h, len(left), left.ctypes.data, right.ctypes.data, prim.ctypes.data, root)

    def bvh(self):
        n = lib().orc_scene_node_count(self.h)
        left, right, prim = (np.zeros(n, np.int32) for _ in range(3))
        boxes = np.zeros((n, 6), np.float32)
        cnt = lib().orc_scene_get_bvh(self.h, left.ctypes.data, right.ctypes.data, prim.ctypes.data, fptr(boxes))
        assert cnt == n
        return left, right, prim, boxes

    def tri_records(self):
        out = np.zeros((self.n_tris, 12), np.float32)
        lib().orc_scene_get_tris(self.h, fptr(out))
        return out

    def render(self, cam, width, height, spp, bounce, tx=28, ty=16, bx=None, by=None, offx=0, offy=0, seed=1984,
               states=None, block_lo=0, block_stride=1, threads=8):
        if bx is None:
            bx, by = width // tx + 1, height // ty + 1
        n = tx * ty * bx * by
        planes = [np.zeros(n, np.float32) for _ in range(9)]
        st = Stats()
        ocam = CameraData.from_buffer_copy(bytes(cam))
        rc = lib().orc_render(self.h, C.byref(ocam), spp, bounce, tx, ty, bx, by, width, height, offx, offy, seed,
                              1 if states is None else 0, None if states is None else states.ctypes.data, block_lo, block_stride,
                              *[fptr(p) for p in planes], threads, C.byref(st))
        assert rc == 0, rc
        return dict(fb=tuple(planes[0:3]), lin=tuple(planes[3:6]), xyz=tuple(planes[6:9]),
                    stats={k: getattr(st, k) for k in ("rays", "paths", "trav_iters", "box_tests", "tri_tests", "max_stack")},
                    geom=dict(tx=tx, ty=ty, bx=bx, by=by, n_lanes=n))

    def trace(self, o, d):
        out = np.zeros(9, np.float32)
        h = lib().orc_trace_ray(self.h, f3(o), f3(d), fptr(out))
        return h, out


def unswizzle(src, tx, ty, bx, by, n_cols, n_rows, offx, offy, image_width, image_height):
    dst = np.zeros(image_width * image_height, np.float32)
    lib().orc_unswizzle(fptr(np.ascontiguousarray(src, np.float32)), fptr(dst), tx, ty, bx, by, n_cols, n_rows, offx, offy, image_width)
    return dst
