"""Host-side scene handle (srt_scene): replaces scene_manager's device-heap world (scene/scene.cuh:103-176)."""
import ctypes as C

import numpy as np

from . import binding as B


def camera_init(width, height, vfov, lookfrom, lookat, vup=(0, 1, 0), defocus_angle=0.0, focus_dist=10.0):
    """camera::initialize (rendering/camera.cu:7-58) -> camera_data (rendering/rendering.cuh:28-36)."""
    cam = B.CameraData()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    B.check(B.lib().srt_camera_init(int(width), int(height), float(vfov), f3(lookfrom), f3(lookat), f3(vup),
                                    float(defocus_angle), float(focus_dist), C.byref(cam)))
    return cam


class Scene:
    def __init__(self, handle):
        if not handle:
            raise B.SrtError(-1, (B.lib().srt_last_error(None) or b"").decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def builtin(cls, scene_id, seed=0):
        return cls(B.lib().srt_scene_builtin(int(scene_id), int(seed)))

    @classmethod
    def from_arrays(cls, tris, materials, background):
        """tris: ctypes array of TriIn; materials: ctypes array of Material; background: 95 floats."""
        s = cls(B.lib().srt_scene_create())
        B.check(B.lib().srt_scene_set_triangles(s._h, tris, len(tris)))
        B.check(B.lib().srt_scene_set_materials(s._h, materials, len(materials)))
        bg = np.ascontiguousarray(background, dtype=np.float32)
        assert bg.shape == (B.N_CIE,)
        B.check(B.lib().srt_scene_set_background(s._h, B.fptr(bg)))
        return s

    def close(self):
        if self._h:
            B.lib().srt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def n_tris(self):
        return B.lib().srt_scene_tri_count(self._h)

    @property
    def n_materials(self):
        return B.lib().srt_scene_material_count(self._h)

    def triangles(self):
        arr = (B.TriIn * self.n_tris)()
        B.check(B.lib().srt_scene_get_triangles(self._h, arr))
        return arr

    def materials(self):
        arr = (B.Material * self.n_materials)()
        B.check(B.lib().srt_scene_get_materials(self._h, arr))
        return arr

    def background(self):
        bg = np.zeros(B.N_CIE, np.float32)
        B.check(B.lib().srt_scene_get_background(self._h, B.fptr(bg)))
        return bg

    def tri_records(self):
        out = np.zeros((self.n_tris, 12), np.float32)
        B.check(B.lib().srt_scene_get_tri_records(self._h, B.fptr(out)))
        return out

    def build_bvh(self, mode=B.BVH_REFERENCE, seed=1984):
        B.check(B.lib().srt_scene_build_bvh(self._h, int(mode), int(seed)))
        return self

    def optimise_bvh(self, passes=3):
        """insertion-based topology optimisation of the built tree (srt_scene_optimise_bvh): for throughput-bound launches"""
        B.check(B.lib().srt_scene_optimise_bvh(self._h, passes))
        return self

    def order_children(self, eye):
        """Re-order every node's children for a viewpoint (nearer child first); upload the scene again afterwards."""
        e = (C.c_float * 3)(*[float(x) for x in eye])
        B.check(B.lib().srt_scene_order_children(self._h, e))
        return self

    @property
    def n_nodes(self):
        return B.lib().srt_scene_node_count(self._h)

    @property
    def is_paired(self):
        """every internal node has two leaf children or none (SAH builder, even triangle count): the launch uses the PAIRED kernel variant"""
        return bool(B.lib().srt_scene_is_paired(self._h))

    @property
    def bvh_depth(self):
        return B.lib().srt_scene_bvh_depth(self._h)

    def bvh(self):
        n = self.n_nodes
        left, right, prim = (np.zeros(n, np.int32) for _ in range(3))
        boxes = np.zeros((n, 6), np.float32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        B.check(B.lib().srt_scene_get_bvh(self._h, ip(left), ip(right), ip(prim), B.fptr(boxes)))
        return left, right, prim, boxes

    def default_camera(self, width, height):
        cam = B.CameraData()
        B.check(B.lib().srt_scene_default_camera(self._h, int(width), int(height), C.byref(cam)))
        return cam
