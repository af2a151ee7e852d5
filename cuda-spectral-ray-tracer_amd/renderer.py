"""Device renderer handle (srt_ctx): the `renderer` / render_manager::step pair of the reference
(rendering/rendering.cuh:39-155, rendering/render_manager.cu:3-66) over the C-ABI.  All compute happens in
libsrt_hip.so on the GPU; this module only moves pointers."""
import ctypes as C

import numpy as np

from . import binding as B

DEFAULT_TX, DEFAULT_TY = 28, 16   # render_manager.cu:93-94


def reference_grid(chunk_w, chunk_h, tx=DEFAULT_TX, ty=DEFAULT_TY):
    """blocks = (w/28+1, h/16+1), render_manager.cu:96"""
    return chunk_w // tx + 1, chunk_h // ty + 1


class Renderer:
    def __init__(self, device=0, _borrowed=None):
        if _borrowed is not None:            # a context owned by a communicator (srt_comm_init_all)
            self._h, self._owned = C.c_void_p(_borrowed), False
            self.device = B.lib().srt_ctx_device(self._h)
        else:
            h = C.c_void_p()
            B.check(B.lib().srt_create(int(device), C.byref(h)))
            self._h, self._owned = h, True
            self.device = device
        self.geom = None
        self.gather_planes = 3

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                B.lib().srt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, code):
        return B.check(code, self._h)

    def upload_scene(self, scene):
        self._ck(B.lib().srt_upload_scene(self._h, scene.handle))

    def launch_plan(self):
        """dict(waves_per_cu, n_cached, all_cached, narrow_refs, paired) of the uploaded scene's render launch"""
        w, n, a, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(B.lib().srt_launch_plan(self._h, C.byref(w), C.byref(n), C.byref(a), C.byref(r)))
        pr = C.c_int()
        self._ck(B.lib().srt_launch_paired(self._h, C.byref(pr)))
        return dict(waves_per_cu=w.value, n_cached=n.value, all_cached=bool(a.value), narrow_refs=bool(r.value), paired=bool(pr.value), test_knobs=self.test_knobs())

    def set_test_knobs(self, wide_refs=False, lds_cache_max=-1, lane_limit=0):
        """tests / tools only (srt_c_api.h): force kernel variants the plan would not pick; upload the scene again afterwards"""
        self._ck(B.lib().srt_set_test_knobs(self._h, 1 if wide_refs else 0, int(lds_cache_max), int(lane_limit)))

    def test_knobs(self):
        w, m, l, e = C.c_int(), C.c_int(), C.c_uint32(), C.c_int()
        self._ck(B.lib().srt_get_test_knobs(self._h, C.byref(w), C.byref(m), C.byref(l), C.byref(e)))
        return dict(wide_refs=bool(w.value), lds_cache_max=m.value, lane_limit=l.value, from_env=bool(e.value))

    def set_camera(self, cam):
        self._ck(B.lib().srt_set_camera(self._h, C.byref(cam)))

    def init_device_params(self, chunk_w, chunk_h, spp, bounce_limit, seed=1984, tx=DEFAULT_TX, ty=DEFAULT_TY, bx=None, by=None):
        if bx is None or by is None:
            bx, by = reference_grid(chunk_w, chunk_h, tx, ty)
        self._ck(B.lib().srt_init_device_params(self._h, tx, ty, bx, by, chunk_w, chunk_h, spp, bounce_limit, seed))
        self.geom = dict(tx=tx, ty=ty, bx=bx, by=by, chunk_w=chunk_w, chunk_h=chunk_h, n_lanes=tx * ty * bx * by)

    def set_partition(self, rank, world):
        self._ck(B.lib().srt_set_partition(self._h, rank, world))

    def set_count_traversal(self, on):
        self._ck(B.lib().srt_set_count_traversal(self._h, 1 if on else 0))

    def render_chunk(self, width, height, offx=0, offy=0, stream=None):
        self._ck(B.lib().srt_render_chunk(self._h, width, height, offx, offy, C.c_void_p(stream or 0)))

    def synchronize(self):
        self._ck(B.lib().srt_synchronize(self._h))

    def set_gather_planes(self, planes):
        """3 (default): the exchange unit is the quantised framebuffer; 9: + the parity planes (unquantised sRGB, XYZ sums)"""
        self._ck(B.lib().srt_set_gather_planes(self._h, planes))
        self.gather_planes = planes

    def tile_buffer(self):
        ptr, n, tl, tp = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint32()
        self._ck(B.lib().srt_tile_buffer(self._h, C.byref(ptr), C.byref(n), C.byref(tl), C.byref(tp)))
        return ptr.value, n.value, tl.value, tp.value

    def copy_tile_buffer(self, dst_ptr, stream=None):
        self._ck(B.lib().srt_copy_tile_buffer(self._h, C.c_void_p(dst_ptr), C.c_void_p(stream or 0)))

    def scatter_tiles(self, gathered_ptr=None, stream=None):
        self._ck(B.lib().srt_scatter_tiles(self._h, C.c_void_p(gathered_ptr or 0), C.c_void_p(stream or 0)))

    def dev_fb(self):
        r, g, b, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._ck(B.lib().srt_dev_fb(self._h, C.byref(r), C.byref(g), C.byref(b), C.byref(n)))
        return r.value, g.value, b.value, n.value

    def read_fb(self):
        n = self.geom["n_lanes"]
        r, g, b = (np.zeros(n, np.float32) for _ in range(3))
        self._ck(B.lib().srt_read_fb(self._h, B.fptr(r), B.fptr(g), B.fptr(b)))
        return r, g, b

    def read_fb_aux(self, which):
        n = self.geom["n_lanes"]
        r, g, b = (np.zeros(n, np.float32) for _ in range(3))
        self._ck(B.lib().srt_read_fb_aux(self._h, which, B.fptr(r), B.fptr(g), B.fptr(b)))
        return r, g, b

    def read_fb_rowmajor(self, image_width, image_height, into=None):
        if into is None:
            into = tuple(np.zeros(image_width * image_height, np.float32) for _ in range(3))
        r, g, b = into
        self._ck(B.lib().srt_read_fb_rowmajor(self._h, B.fptr(r), B.fptr(g), B.fptr(b), image_width, image_height))
        return r, g, b

    def tile_costs(self, with_max_pixel=False):
        """the cost probe's node visits per local tile; with_max_pixel: (per tile, of each tile's most expensive pixel)"""
        _, _, tl, _ = self.tile_buffer()
        out = np.zeros(tl * (2 if with_max_pixel else 1), np.uint32)
        self._ck(B.lib().srt_get_tile_costs(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return (out[:tl], out[tl:]) if with_max_pixel else out

    def order_children_by_profile(self, scene, width, height, spp, bounce_limit, min_samples=4):
        """re-order the children of `scene`'s tree from one instrumented probe frame of this context's camera (srt_c_api.h); the
        scene is left uploaded; returns the number of nodes whose children were swapped.  init_device_params comes next."""
        n = C.c_uint32(0)
        self._ck(B.lib().srt_order_children_by_profile(self._h, scene.handle, width, height, spp, bounce_limit, min_samples, C.byref(n)))
        return n.value

    def stats(self):
        st = B.Stats()
        self._ck(B.lib().srt_get_stats(self._h, C.byref(st)))
        d = {k: getattr(st, k) for k in ("rays", "paths", "node_visits", "tri_tests", "box_tests")}
        d["util"] = list(st.util)
        d["shade"] = list(st.shade)
        d["waves"] = list(st.waves)
        d["hits"] = st.hits
        d["max_pixel_node_visits"], d["max_pixel_rays"] = st.reserved[0], st.reserved[1]
        return d

    def wave_debug(self):
        """per persistent wave of the last INSTRUMENTED launch: [life, queue-dry time (256-cycle units), rays, rays of its most expensive pixel]"""
        n = int(self.stats()["waves"][0])
        out = np.zeros((n, 4), np.uint32)
        if n:
            self._ck(B.lib().srt_get_wave_debug(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def last_kernel_ms(self):
        ms = C.c_float()
        self._ck(B.lib().srt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def calibrate(self, kind, waves_per_simd=4, iters=20000):
        """Issue-rate microkernel (csrc/srt_calib.hip): one workgroup of waves_per_simd * 256 threads on every CU.

        The SIMD's arbiter prefers older waves, so with four resident waves two of them run at their single-wave speed and
        finish at about half of the launch; the rate a SIMD sustains is therefore the instructions of ALL its waves over the
        cycles of the LAST one (wave_cycles_max, = wall x clock), not over the mean wave life -- the mean-based figure of round 2
        (0.585 / cycle) belongs to no SIMD.  `instr_per_cycle_per_simd` is the max-based rate; min / mean / max are all returned."""
        cal = B.Calibration()
        self._ck(B.lib().srt_calibrate(self._h, kind, waves_per_simd, iters, C.byref(cal)))
        n_simd = cal.n_cu * 4
        total = cal.instr_per_wave * cal.n_waves
        clock_ghz = cal.wave_cycles_max / (cal.wall_ms * 1e-3) / 1e9        # the slowest wave spans the launch: its cycles / wall
        return dict(kind=kind, waves_per_simd=cal.waves_per_simd, n_cu=cal.n_cu, wave_cycles_min=cal.wave_cycles_min,
                    wave_cycles_mean=cal.wave_cycles_mean, wave_cycles_max=cal.wave_cycles_max, wall_ms=cal.wall_ms,
                    instr_per_wave=cal.instr_per_wave,
                    instr_per_cycle_per_simd=cal.instr_per_wave * cal.waves_per_simd / cal.wave_cycles_max,
                    instr_per_cycle_per_simd_mean_wave=cal.instr_per_wave * cal.waves_per_simd / cal.wave_cycles_mean,
                    instr_per_s=total / (cal.wall_ms * 1e-3), clock_ghz=clock_ghz, n_simd=n_simd)

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 4), np.float32)
        self._ck(B.lib().srt_trace_rays(self._h, B.fptr(rays), rays.shape[0], B.fptr(out)))
        return out

    def op_sweep(self, which, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        out = np.zeros_like(a)
        self._ck(B.lib().srt_device_op_sweep(self._h, which, B.fptr(a), B.fptr(b), a.size, B.fptr(out)))
        return out


class Comm:
    """Multi-GPU communicator (srt_comm): W ranks render interleaved 8x8 tiles of a chunk, one RCCL gather to rank 0.
    Comm.init_all(devices): one process drives several GPUs.  Comm.init_rank(renderer, id, rank, world): one process per
    GPU, `id` from Comm.unique_id() on rank 0."""

    def __init__(self, handle, renderers, owns):
        self._h, self.renderers, self._owns = handle, renderers, owns
        self.world = B.lib().srt_comm_world(handle)

    @staticmethod
    def available():
        """True when an RCCL can be loaded in this process (no collective involved)"""
        return B.lib().srt_comm_available() == 0

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * B.COMM_ID_BYTES)()
        B.check_comm(B.lib().srt_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def init_all(cls, devices):
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        B.check_comm(B.lib().srt_comm_init_all(arr, len(devices), C.byref(h)))
        rs = [Renderer(_borrowed=B.lib().srt_comm_ctx(h, k)) for k in range(B.lib().srt_comm_local_count(h))]
        return cls(h, rs, True)

    @classmethod
    def init_rank(cls, renderer, comm_id, rank, world):
        buf = (C.c_ubyte * B.COMM_ID_BYTES).from_buffer_copy(comm_id)
        h = C.c_void_p()
        B.check_comm(B.lib().srt_comm_init_rank(renderer._h, buf, rank, world, C.byref(h)))
        return cls(h, [renderer], False)

    def close(self):
        if getattr(self, "_h", None):
            B.lib().srt_comm_destroy(self._h)
            self._h = None
            for r in self.renderers:
                if not r._owned:
                    r._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, code):
        return B.check_comm(code, self._h)

    @property
    def root(self):
        """rank 0's renderer (holds the assembled framebuffer), None when rank 0 lives in another process"""
        p = B.lib().srt_comm_root_ctx(self._h)
        for r in self.renderers:
            if r._h and r._h.value == p:
                return r
        return None

    def upload_scene(self, scene):
        self._ck(B.lib().srt_comm_upload_scene(self._h, scene.handle))

    def set_camera(self, cam):
        self._ck(B.lib().srt_comm_set_camera(self._h, C.byref(cam)))

    def init_device_params(self, chunk_w, chunk_h, spp, bounce_limit, seed=1984, tx=DEFAULT_TX, ty=DEFAULT_TY, bx=None, by=None):
        if bx is None or by is None:
            bx, by = reference_grid(chunk_w, chunk_h, tx, ty)
        self._ck(B.lib().srt_comm_init_device_params(self._h, tx, ty, bx, by, chunk_w, chunk_h, spp, bounce_limit, seed))
        for r in self.renderers:
            r.geom = dict(tx=tx, ty=ty, bx=bx, by=by, chunk_w=chunk_w, chunk_h=chunk_h, n_lanes=tx * ty * bx * by)

    def set_gather_planes(self, planes):
        self._ck(B.lib().srt_comm_set_gather_planes(self._h, planes))
        for r in self.renderers:
            r.gather_planes = planes

    def last_gather_ms(self):
        ms = C.c_float()
        self._ck(B.lib().srt_comm_last_gather_ms(self._h, C.byref(ms)))
        return ms.value

    def render_frame(self, width, height, offx=0, offy=0):
        self._ck(B.lib().srt_render_frame_multi(self._h, width, height, offx, offy))

    def synchronize(self):
        self._ck(B.lib().srt_comm_synchronize(self._h))

    def stats(self):
        rays, paths, ms = C.c_uint64(), C.c_uint64(), C.c_float()
        self._ck(B.lib().srt_comm_stats(self._h, C.byref(rays), C.byref(paths), C.byref(ms)))
        return dict(rays=rays.value, paths=paths.value, max_kernel_ms=ms.value)


def pixels_per_lane(renderer, width, height, world=1):
    """pixels of a width x height frame per persistent lane of one rank's launch (CUs x waves per CU x 64 lanes; scene uploaded):
    below about 6 a launch is bound by its longest pixel chain, above by total work"""
    lanes = B.lib().srt_ctx_cu_count(renderer._h) * renderer.launch_plan()["waves_per_cu"] * 64
    return width * height / float(max(world, 1)) / max(lanes, 1)


def tune_tree_for_throughput(renderer, scene, width, height, bounce_limit):
    """Tree preparation for a THROUGHPUT-bound launch of a width x height frame on this build's own SAH tree: insertion-based topology
    optimisation (trees of up to 8 192 triangles: measured no gain above) and the child order from a probe frame.  Not for launches
    with few pixels per lane (see pixels_per_lane): less total work is not a cheaper longest pixel -- measured 5-10 % slower there.
    Returns a description for the record.  Deterministic: every rank arrives at the same tree."""
    notes = []
    if scene.n_tris <= 8192:
        renderer.upload_scene(scene)
        resident = renderer.launch_plan()["all_cached"]
        scene.optimise_bvh(3)
        renderer.upload_scene(scene)
        if resident and not renderer.launch_plan()["all_cached"]:
            # (reinsertion may deepen the tree: deeper LDS stacks, fewer cached records -- a tree that just fitted LDS no longer does,
            # which costs far more than the passes return)
            scene.build_bvh(B.BVH_SAH, 1984)
            notes.append("reinsertion undone (the deeper tree would no longer be LDS resident)")
        else:
            notes.append("3 reinsertion passes")
    n, (pw, ph, ps) = profile_child_order(renderer, scene, width, height, bounce_limit)
    notes.append(("child order profiled on a %dx%d x %d spp probe frame: %d nodes swapped" % (pw, ph, ps, n)) if n else
                 ("builder's child order kept (the %dx%d x %d spp probe frame was not cheaper with the profiled one)" % (pw, ph, ps)))
    return "; ".join(notes)


def profile_child_order(renderer, scene, width, height, bounce_limit):
    """The standard use of srt_order_children_by_profile for a frame of width x height: probe frame at a quarter of the size, 8 spp,
    nodes with at least 16 deciding rays, from the scene's default camera.  Returns the number of nodes whose children were swapped
    (0: the probe frame was not cheaper with the profiled order and the builder's order was kept).  Deterministic."""
    pw, ph = max(width // 4, 32), max(height // 4, 32)
    renderer.set_camera(scene.default_camera(pw, ph))
    return renderer.order_children_by_profile(scene, pw, ph, 8, bounce_limit, 16), (pw, ph, 8)


def render_image(scene, cam, width, height, spp, bounce_limit, seed=1984, device=0, count_traversal=False, renderer=None):
    """Whole-image single-chunk render on one GPU (the reference's default configuration, Q13).
    Returns dict with block-linear planes, row-major quantised planes, stats and kernel ms."""
    r = renderer or Renderer(device)
    r.upload_scene(scene)
    r.set_camera(cam)
    r.init_device_params(width, height, spp, bounce_limit, seed)
    r.set_partition(0, 1)
    r.set_count_traversal(count_traversal)
    planes_before = r.gather_planes
    r.set_gather_planes(9)            # the parity planes (unquantised sRGB, XYZ sums) are part of what this returns
    r.render_chunk(width, height, 0, 0)
    r.scatter_tiles()
    r.set_gather_planes(planes_before)
    out = dict(fb=r.read_fb(), lin=r.read_fb_aux(1), xyz=r.read_fb_aux(2), rowmajor=r.read_fb_rowmajor(width, height),
               stats=r.stats(), kernel_ms=r.last_kernel_ms(), geom=dict(r.geom))
    if renderer is None:
        r.close()
    return out
