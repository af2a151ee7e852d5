"""ctypes binding of the C-ABI declared in include/srt_c_api.h (libsrt_hip.so).

Nothing in here computes: it declares the structs / prototypes and turns negative status codes into
exceptions.  There is no fallback path: if the shared library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRT_LIB_PATH") or os.path.join(_HERE, "libsrt_hip.so")   # (override: kernel experiments only)

N_CIE = 95
TILE_PLANES = 9
TILE_LANES = 64

SCENE_CORNELL, SCENE_PRISM, SCENE_TRIS, SCENE_RANDOM_SPHERES, SCENE_MESH100K = 0, 1, 2, 100, 101
BVH_REFERENCE, BVH_SAH = 0, 1
MAT_LAMBERTIAN, MAT_METALLIC, MAT_DIELECTRIC, MAT_EMISSIVE, MAT_NO_MAT = 0, 1, 2, 4, 6


class TriIn(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3),
                ("mat_index", C.c_uint32), ("aa_plane", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("col", C.c_float * 3), ("reflection_fuzz", C.c_float), ("material_type", C.c_uint32),
                ("spectral_distribution", C.c_float * N_CIE), ("emission_power", C.c_float),
                ("sellmeier_B", C.c_float * 3), ("sellmeier_C", C.c_float * 3)]


class CameraData(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3), ("pixel00_loc", C.c_float * 3),
                ("defocus_angle", C.c_float),
                ("camera_center", C.c_float * 3), ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("box_tests", C.c_uint64), ("util", C.c_uint64 * 9), ("reserved", C.c_uint64 * 2), ("shade", C.c_uint64 * 4),
                ("waves", C.c_uint64 * 4), ("hits", C.c_uint64)]


class Calibration(C.Structure):
    _fields_ = [("wave_cycles_mean", C.c_double), ("wave_cycles_max", C.c_double), ("wall_ms", C.c_double),
                ("instr_per_wave", C.c_uint64), ("n_waves", C.c_uint32), ("n_cu", C.c_uint32), ("waves_per_simd", C.c_uint32),
                ("reserved", C.c_uint32), ("wave_cycles_min", C.c_double)]


assert C.sizeof(Material) == 428 and C.sizeof(CameraData) == 84 and C.sizeof(TriIn) == 44

# every symbol include/srt_c_api.h declares: name -> (restype, argtypes)
_vp, _i, _u32, _u64, _f, _sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)
PROTOTYPES = {
    "srt_version": (C.c_char_p, []),
    "srt_camera_init": (_i, [_i, _i, _f, _fp, _fp, _fp, _f, _f, C.POINTER(CameraData)]),
    "srt_scene_create": (_vp, []),
    "srt_scene_builtin": (_vp, [_i, _u64]),
    "srt_scene_destroy": (None, [_vp]),
    "srt_scene_default_camera": (_i, [_vp, _i, _i, C.POINTER(CameraData)]),
    "srt_scene_set_triangles": (_i, [_vp, C.POINTER(TriIn), _sz]),
    "srt_scene_set_materials": (_i, [_vp, C.POINTER(Material), _sz]),
    "srt_scene_set_background": (_i, [_vp, _fp]),
    "srt_scene_tri_count": (_sz, [_vp]),
    "srt_scene_material_count": (_sz, [_vp]),
    "srt_scene_get_triangles": (_i, [_vp, C.POINTER(TriIn)]),
    "srt_scene_get_materials": (_i, [_vp, C.POINTER(Material)]),
    "srt_scene_get_background": (_i, [_vp, _fp]),
    "srt_scene_get_tri_records": (_i, [_vp, _fp]),
    "srt_material_bake": (_i, [C.POINTER(Material)]),
    "srt_set_reference_quirks": (_i, [_i]),
    "srt_bake_sigmoid_spectrum": (_i, [_fp, _f, _i, _fp]),
    "srt_fit_sigmoid_coeffs": (_i, [_fp, _fp]),
    "srt_color_tables": (_i, [_fp, _fp]),
    "srt_background_spectrum": (_i, [_fp, _fp]),
    "srt_rotation_matrix": (_i, [_f, _i, _fp]),
    "srt_scene_build_bvh": (_i, [_vp, _i, _u64]),
    "srt_scene_order_children": (_i, [_vp, _fp]),
    "srt_scene_optimise_bvh": (_i, [_vp, C.c_int]),
    "srt_scene_is_paired": (_i, [_vp]),
    "srt_scene_node_count": (_sz, [_vp]),
    "srt_scene_bvh_depth": (_i, [_vp]),
    "srt_scene_get_bvh": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp]),
    "srt_create": (_i, [_i, C.POINTER(_vp)]),
    "srt_destroy": (None, [_vp]),
    "srt_last_error": (C.c_char_p, [_vp]),
    "srt_upload_scene": (_i, [_vp, _vp]),
    "srt_set_camera": (_i, [_vp, C.POINTER(CameraData)]),
    "srt_launch_plan": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "srt_launch_paired": (_i, [_vp, C.POINTER(_i)]),
    "srt_set_test_knobs": (_i, [_vp, _i, _i, _u32]),
    "srt_get_test_knobs": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_u32), C.POINTER(_i)]),
    "srt_launch_lds_bytes": (_i, [_vp, C.POINTER(_sz)]),
    "srt_init_device_params": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u64]),
    "srt_set_partition": (_i, [_vp, _u32, _u32]),
    "srt_render_chunk": (_i, [_vp, _u32, _u32, _u32, _u32, _vp]),
    "srt_synchronize": (_i, [_vp]),
    "srt_set_gather_planes": (_i, [_vp, _u32]),
    "srt_tile_buffer": (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_u32), C.POINTER(_u32)]),
    "srt_copy_tile_buffer": (_i, [_vp, _vp, _vp]),
    "srt_scatter_tiles": (_i, [_vp, _vp, _vp]),
    "srt_dev_fb": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz)]),
    "srt_read_fb": (_i, [_vp, _fp, _fp, _fp]),
    "srt_read_fb_rowmajor": (_i, [_vp, _fp, _fp, _fp, _u32, _u32]),
    "srt_read_fb_aux": (_i, [_vp, _i, _fp, _fp, _fp]),
    "srt_get_tile_costs": (_i, [_vp, C.POINTER(C.c_uint32), _sz]),
    "srt_order_children_by_profile": (_i, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "srt_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "srt_set_count_traversal": (_i, [_vp, _i]),
    "srt_get_wave_debug": (_i, [_vp, C.POINTER(C.c_uint32), _sz]),
    "srt_last_kernel_ms": (_i, [_vp, C.POINTER(_f)]),
    "srt_trace_rays": (_i, [_vp, _fp, _sz, _fp]),
    "srt_device_op_sweep": (_i, [_vp, _i, _fp, _fp, _sz, _fp]),
    "srt_calibrate": (_i, [_vp, _i, _u32, _u32, C.POINTER(Calibration)]),
    "srt_ctx_device": (_i, [_vp]),
    "srt_ctx_cu_count": (_i, [_vp]),
    "srt_comm_init_all": (_i, [C.POINTER(_i), _i, C.POINTER(_vp)]),
    "srt_comm_unique_id": (_i, [C.POINTER(C.c_ubyte)]),
    "srt_comm_init_rank": (_i, [_vp, C.POINTER(C.c_ubyte), _u32, _u32, C.POINTER(_vp)]),
    "srt_comm_available": (_i, []),
    "srt_comm_set_gather_planes": (_i, [_vp, _u32]),
    "srt_comm_last_gather_ms": (_i, [_vp, C.POINTER(_f)]),
    "srt_comm_destroy": (None, [_vp]),
    "srt_comm_last_error": (C.c_char_p, [_vp]),
    "srt_comm_world": (_u32, [_vp]),
    "srt_comm_local_count": (_u32, [_vp]),
    "srt_comm_ctx": (_vp, [_vp, _u32]),
    "srt_comm_root_ctx": (_vp, [_vp]),
    "srt_comm_upload_scene": (_i, [_vp, _vp]),
    "srt_comm_set_camera": (_i, [_vp, C.POINTER(CameraData)]),
    "srt_comm_init_device_params": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u64]),
    "srt_render_frame_multi": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "srt_comm_synchronize": (_i, [_vp]),
    "srt_comm_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_f)]),
}


class SrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("srt error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libsrt_hip.so (built in-tree by __graft_entry__.build()).  No fallback: raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the render path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


COMM_ID_BYTES = 128


def check_comm(code, comm=None):
    if code is not None and code < 0:
        msg = lib().srt_comm_last_error(comm)
        raise SrtError(code, msg.decode() if msg else "")
    return code


def check(code, ctx=None):
    if code is not None and code < 0:
        msg = lib().srt_last_error(ctx)
        raise SrtError(code, msg.decode() if msg else "")
    return code


def fptr(arr):
    """float32 C-contiguous numpy array -> float*"""
    return arr.ctypes.data_as(_fp)
