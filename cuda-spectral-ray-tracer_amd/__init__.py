"""MI355X-native spectral path tracer: hot-path replacement for PieSil/CUDA-spectral-ray-tracer.

The product is libsrt_hip.so (hand-written gfx950 kernels behind the C-ABI of include/srt_c_api.h).  This
package is the Python face of that ABI (ctypes) plus the torch.distributed plumbing of the multi-GPU gather.
Importing it never computes anything and never falls back to a CPU implementation.
"""
from . import binding
from .binding import (BVH_REFERENCE, BVH_SAH, SCENE_CORNELL, SCENE_MESH100K, SCENE_PRISM, SCENE_RANDOM_SPHERES,
                      SCENE_TRIS, CameraData, Material, SrtError, TriIn)
from .scene import Scene, camera_init
from .renderer import Comm, Renderer, pixels_per_lane, profile_child_order, reference_grid, render_image, tune_tree_for_throughput
from . import tiles

__all__ = ["binding", "Scene", "camera_init", "Renderer", "Comm", "reference_grid", "render_image", "profile_child_order", "tune_tree_for_throughput", "pixels_per_lane", "tiles", "SrtError",
           "TriIn", "Material", "CameraData", "BVH_REFERENCE", "BVH_SAH", "SCENE_CORNELL", "SCENE_PRISM",
           "SCENE_TRIS", "SCENE_RANDOM_SPHERES", "SCENE_MESH100K"]
