#!/usr/bin/env python3
"""bench.py -- headline benchmark of the spectral path-tracing hot path on MI355X.

Metric (BASELINE.json): Mray/s on the synthetic "random spheres" scene, 1920x1080, 1024 spp, depth 16
(config[2]; a ray = one closest-hit query).  One *step* = one full render of that image: every rank
renders its interleaved 8x8 tiles with the HIP kernel, ONE gather (RCCL over xGMI, issued by the library:
srt_render_frame_multi) brings the compact tile buffers to rank 0, which scatters them into the
block-linear planar framebuffer in HBM.  Scene, camera and RNG states are resident in HBM before the timed
region.  The image is fixed, so N GPUs split the same work: scaling is "strong".

Three ways to start it (config.launch_mode says which one ran):

  python bench.py --gpus 1 --steps K --warmup W                      one GPU
  python bench.py --gpus N --steps K --warmup W                      ONE process drives N GPUs: srt_comm_init_all (ncclCommInitAll,
                                                                     one HIP stream per device) -- SURVEY 8(e)'s shape
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                      one process per GPU: srt_comm_init_rank; torch.distributed carries
                                                                     the 128-byte communicator id, the barriers and the statistics

Wall-clock budget of one invocation (the driver allows 600 s): import torch on a fresh box <= 120 s; scene build + tree tuning <= 3 s;
communicator + verification frame <= 30 s; (W + K) headline frames of <= 0.36 s each at N = 1 (less at N > 1); one builder's-tree frame;
N = 1 only: CPU baseline ~ 35 s and the three other configurations ~ 3 s; the cfg 5 sub-record 12.8 s / N (+ 1 s build).  Whatever
happened before it (a communicator that fell back, a slow box), the cfg 5 sub-record is SKIPPED -- and the line says so -- when its
estimated cost would carry the process past --time-budget seconds (default 420) since it started: the headline line is always printed.

Rank 0 prints ONE JSON line.  Besides the headline it carries
  cfg5            one frame of BASELINE cfg 5 as specified (100k-triangle mesh, 3840x2160, 4096 spp) on the same N GPUs,
  other_configs   (N = 1) cfg 2, cfg 4 and cfg 5's scene at 512 spp, each with V, T, kernel time and its own roofline entry,
  roofline        the render kernel against what binds it (see below), cpu_baseline (N = 1) the CPU oracle on this box's host cores.

`roofline` prices the render kernel against what actually binds it.  The scene is LDS / L2 resident, so HBM is
not the roof (the SURVEY 8(d) HBM figure is kept under roofline.hbm, against the copy bandwidth measured in this
run; a value above 1 there says the bytes are served by LDS / L2).  The kernel is a divergent VALU program; the roof is
the rate at which the chip issues wave64 vector instructions.  Two peaks are reported:
  frac       against the rate MEASURED in this run by the library's calibration microkernel (srt_calibrate,
             csrc/srt_calib.hip: independent v_add_f32, 4 waves / SIMD, every CU; wall-clock based) x 64 lanes;
  frac_arch  against the architectural 256 CU x 4 SIMD x 32 lanes / clk x 2.4 GHz = 78.6 T lane-op/s
             (MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD).
achieved = useful VALU lane-ops per ray (active lanes summed over every vector instruction: SQ_THREAD_CYCLES_VALU / rays
from a committed rocprofv3 --pmc pass, profiles/rNN/lane_ops_per_ray.json -- IMPORTED, labelled so, and tied to the kernel
binary by the sha256 of the kernel's machine code in the gfx950 code object of the library that is loaded (tools/kernel_id.py): a
mismatch is reported as "achieved_source": "STALE ...") x rays per launch (live) / render-kernel time (live, HIP events on the
launch stream).

`cpu_baseline` is the CPU oracle (a port: the reference has no CPU path) timed on this box's USABLE host cores
(scheduler affinity and cgroup quota, not os.cpu_count()) on a bounded sample of the same workload, its one-thread rate
on the same scene, and BASELINE.md's "config 1, one thread, full size" leg.
"""
import argparse
import json
import os
import sys
import time

# Read by the HSA runtime when it STARTS (the first torch.cuda call / HIP call of the process): must be in the environment before
# `import torch` below in main().  The host driver of this pool only supports dmabuf IPC; without it RCCL's peer-memory exchange
# between processes fails with `hipIpcGetMemHandle: invalid argument`.  tests/test_bench_helpers.py checks the order.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_PROCESS_START = time.time()

HBM_SPEC_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy rate is measured below
NODE_BYTES, TRI_BYTES, MAT_BYTES = 64, 48, 56                  # SURVEY 8(d)'s formula: V * 64 + T * 48 + 56
INNER_BYTES, FRINGE_BYTES, SHADE_BYTES = 64, 96, 48            # this build's records: INNER 64 B in HBM (52 B LDS image), FRINGE 96 B (both children), shading 48 B per hit
LANE_OPS_FILES = [os.path.join(ROOT, "profiles", r, "lane_ops_per_ray.json") for r in ("r05", "r04", "r03", "r02")]      # newest first
ARCH_PEAK_GLANEOPS = 256 * 4 * 32 * 2.4      # G lane-op/s: 256 CU x 4 SIMD x 32 lanes per clock x 2.4 GHz (157.3 TFLOP/s fp32 FMA / 2)
SCENE_NAMES = {0: "reference CORNELL scene", 1: "reference PRISM scene", 2: "reference TRIS scene", 100: "random-spheres tri scene",
               101: "100k-triangle mesh in the Cornell shell"}
BVH_NAMES = {1: "SAH (exact sweep below 8192 triangles, 256 bins above), one triangle per leaf, children ordered by distance to the scene's default camera",
             0: "reference builder (bvh/bvh.cu:206-346)"}


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def oracle_scene(srt, scene, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as O
    osc = O.OracleScene(scene.triangles(), scene.materials(), scene.background())
    if mode == srt.BVH_REFERENCE:
        assert osc.build_reference(1984) == 1
    else:
        left, right, prim, _ = scene.bvh()
        assert osc.set_bvh(left, right, prim, 0) == 1
    return osc


def usable_cores():
    """Host cores this process may really use: scheduler affinity, capped by a cgroup CPU quota when one is set.
    (os.cpu_count() reports the machine -- 256 on the GPU boxes -- whatever share the container was given.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        n = os.cpu_count() or 1
    note = "sched_getaffinity: %d (os.cpu_count: %s)" % (n, os.cpu_count())
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:      # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:      # noqa: BLE001
            pass
    if quota is not None:
        note += ", cgroup quota %.1f CPUs" % quota
        n = max(1, min(n, int(quota + 0.999)))
    return n, note


def cpu_baseline(srt, scene, cam, width, height, depth, mode, budget_s=18.0, gpu_renderer=None):
    """The oracle (kind "port") on the usable host cores, same scene/camera/size, reduced spp (the rate is spp independent),
    next to its ONE-thread rate on the same scene."""
    cores, cores_note = usable_cores()
    osc = oracle_scene(srt, scene, mode)
    # one thread, same scene: a half-resolution frame at 1 spp (same camera, so the same mix of sky / geometry / glass paths)
    t0 = time.time()
    r1 = osc.render(scene.default_camera(width // 2, height // 2), width // 2, height // 2, 1, depth, threads=1)
    dt_one = max(time.time() - t0, 1e-6)
    one_thread = {"value": r1["stats"]["rays"] / dt_one / 1e6, "unit": "Mray/s", "cores": 1,
                  "sample": "%dx%d, 1 spp, depth %d, same scene / BVH, 1 thread, %.1f s" % (width // 2, height // 2, depth, dt_one)}
    t0 = time.time()
    r = osc.render(cam, width, height, 1, depth, threads=cores)           # calibration pass: 1 spp
    dt1 = max(time.time() - t0, 1e-3)
    spp = int(max(1, min(64, budget_s / dt1)))
    t0 = time.time()
    r = osc.render(cam, width, height, spp, depth, threads=cores)
    dt = time.time() - t0
    st = r["stats"]
    out = {"value": st["rays"] / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
           "sample": "%dx%d, %d spp, depth %d, same scene/camera/BVH, %d threads, %.1f s" % (width, height, spp, depth, cores, dt),
           "cores_source": cores_note, "one_thread_same_scene": one_thread,
           "note": "rounds 1-2 started os.cpu_count() = 256 threads on a box that grants this process far fewer cores (see cores_source), "
                   "which is why their '256 threads' figure (10-12 Mray/s) was below this one; 1.2 % of the rays (NaN directions, quirk Q1) walk the whole tree "
                   "on the CPU like in the reference"}
    if gpu_renderer is not None:
        # the same frame at the same spp on the GPU, by BOTH builds of the kernel -- the production one (the kernel this run timed:
        # assembly traversal block) and the instrumented one: per-channel L-inf of the unquantised sRGB planes (BASELINE's parity
        # metric, target <= 1e-3) and the number of lanes whose bits differ -- the checker at work, outside the timed region
        import numpy as np
        out["parity"] = {"spp": spp}
        for name, counted in (("production_kernel", False), ("instrumented_kernel", True)):
            g = srt.render_image(scene, cam, width, height, spp, depth, renderer=gpu_renderer, count_traversal=counted)
            linf = [float(np.max(np.abs(a - b))) for a, b in zip(g["lin"], r["lin"])]
            nbits = int(sum(int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))) for a, b in zip(g["xyz"], r["xyz"])))
            nq = int(sum(int(np.count_nonzero(a != b)) for a, b in zip(g["fb"], r["fb"])))
            out["parity"][name] = {"linf_rgb": linf, "lanes_with_different_bits": nbits, "quantised_values_different": nq,
                                   "rays_equal": bool(g["stats"]["rays"] == st["rays"])}
    return out


def cfg1_single_thread(srt, budget_s=25.0):
    """BASELINE.md CPU-baseline plan item 1: config 1 (reference CORNELL scene, 256x256, 16 spp, depth 8, reference BVH) on ONE
    host thread, full size; wall seconds and Mray/s.  If the full run would not fit the budget it is cut to fewer spp and says so."""
    scene = srt.Scene.builtin(srt.SCENE_CORNELL, 0).build_bvh(srt.BVH_REFERENCE, 1984)
    W = H = 256
    cam = scene.default_camera(W, H)
    osc = oracle_scene(srt, scene, srt.BVH_REFERENCE)
    t0 = time.time()
    osc.render(cam, W, H, 1, 8, threads=1)
    dt1 = max(time.time() - t0, 1e-3)
    spp = 16 if 16 * dt1 <= budget_s else max(1, int(budget_s / dt1))
    t0 = time.time()
    r = osc.render(cam, W, H, spp, 8, threads=1)
    dt = time.time() - t0
    return {"wall_s": dt, "value": r["stats"]["rays"] / dt / 1e6, "unit": "Mray/s", "cores": 1, "kind": "port",
            "sample": "CORNELL 256x256, %d spp%s, depth 8, reference BVH, 1 thread" % (spp, "" if spp == 16 else " (of 16: budget)")}


def measure_hbm_copy_gbs(torch):
    """Device-to-device copy rate of this GPU (the HBM figure is divided by a MEASURED peak): read + write bytes / s."""
    n = 1 << 28                                    # 1 GiB of fp32 per buffer: far beyond the 256 MiB Infinity Cache
    a = torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    del a, b
    return 2 * n * 4 / (ms * 1e-3) / 1e9


def checksum_of(renderer):
    fb = renderer.read_fb()
    return int(sum(int(p.astype("int64").sum()) for p in fb))


# ----------------------------------------------------------------------------------------------------------------------
# The ranks of the job as THIS process sees them.  Three shapes behind one interface:
#   upload(scene, cam) / init_params(W, H, spp, depth) / set_count(on) / frame(W, H) -- one complete frame incl. the exchange,
#   returns when it is done -- / barrier() / local (the renderers this process drives) / root (rank 0's renderer or None)
# ----------------------------------------------------------------------------------------------------------------------
class OneGpu:
    launch_mode = "one process, one GPU"

    def __init__(self, srt, torch, device):
        self.torch, self.world, self.rank = torch, 1, 0
        self.r = srt.Renderer(device)
        self.local, self.root = [self.r], self.r
        self.stream = torch.cuda.current_stream().cuda_stream
        self.gather_via, self.gather_check = "none (1 GPU)", None

    def upload(self, scene, cam):
        self.r.upload_scene(scene); self.r.set_camera(cam); self.r.set_partition(0, 1)

    def init_params(self, W, H, spp, depth):
        self.r.init_device_params(W, H, spp, depth, 1984)

    def set_count(self, on):
        self.r.set_count_traversal(on)

    def frame(self, W, H):
        self.r.render_chunk(W, H, 0, 0, self.stream)
        self.r.scatter_tiles(None, self.stream)
        self.torch.cuda.synchronize()

    def exchange_ms(self):
        return None

    def barrier(self):
        self.torch.cuda.synchronize()

    def reduce_sum(self, values):
        return list(values)

    def reduce_max(self, values):
        return list(values)

    def gather_rows(self, row):
        return [list(row)]

    def close(self):
        self.r.close()


class OneProcessManyGpus:
    """`python bench.py --gpus N` without a launcher: srt_comm_init_all -- one process, N contexts, ncclCommInitAll, one HIP stream per
    device; srt_render_frame_multi enqueues the N render kernels, ONE grouped ncclGather and the scatter.  If the communicator
    cannot be formed, or its verification frame differs from a single-context render, the exchange falls back to device-to-device
    copies through torch (same bytes, same links, no collective) and the JSON says so."""
    launch_mode = "one process drives all GPUs (srt_comm_init_all: ncclCommInitAll, one HIP stream per device)"

    def __init__(self, srt, torch, devices):
        self.srt, self.torch, self.devices = srt, torch, list(devices)
        self.world, self.rank = len(devices), 0
        self.comm, self.gather_check = None, None
        try:
            self.comm = srt.Comm.init_all(self.devices)
            self.local = self.comm.renderers
            self.root = self.comm.root
            self.gather_via = "srt_render_frame_multi (ncclGather inside libsrt_hip.so, communicator from srt_comm_init_all)"
        except Exception as e:      # noqa: BLE001 -- any failure here must not lose the measurement
            log("srt_comm_init_all failed (%r); exchanging the tile buffers with device-to-device copies instead" % (e,))
            self._fallback("library communicator unavailable: %r" % (e,))

    def _fallback(self, why):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.local = [self.srt.Renderer(d) for d in self.devices]
        for k, r in enumerate(self.local):
            r.set_partition(k, self.world)
        self.root = self.local[0]
        self._staging, self._gathered = None, None
        self.gather_via = "device-to-device copies through torch (%s)" % why

    def upload(self, scene, cam):
        self._scene, self._cam = scene, cam
        if self.comm is not None:
            self.comm.upload_scene(scene); self.comm.set_camera(cam)
        else:
            for r in self.local:
                r.upload_scene(scene); r.set_camera(cam)

    def init_params(self, W, H, spp, depth):
        self._depth = depth
        if self.comm is not None:
            self.comm.init_device_params(W, H, spp, depth, 1984)
        else:
            for r in self.local:
                r.init_device_params(W, H, spp, depth, 1984)

    def set_count(self, on):
        for r in self.local:
            r.set_count_traversal(on)

    def frame(self, W, H):
        if self.comm is not None:
            self.comm.render_frame(W, H, 0, 0)
            self.comm.synchronize()
            return
        torch = self.torch
        for r in self.local:
            r.render_chunk(W, H, 0, 0, None)          # asynchronous: the devices render side by side
        _, n_floats, _, _ = self.local[0].tile_buffer()
        if self._staging is None or self._staging[0].numel() != n_floats:
            self._staging = [torch.empty(n_floats, dtype=torch.float32, device="cuda:%d" % d) for d in self.devices]
            self._gathered = torch.empty((self.world, n_floats), dtype=torch.float32, device="cuda:%d" % self.devices[0])
        for r, t in zip(self.local, self._staging):
            r.copy_tile_buffer(t.data_ptr(), None)
            r.synchronize()
        for k, t in enumerate(self._staging):
            self._gathered[k].copy_(t)
        torch.cuda.synchronize(self.devices[0])
        self.root.scatter_tiles(self._gathered.data_ptr(), None)
        self.root.synchronize()

    def verify(self, W, H, depth, spp=8):
        """one cheap frame through the communicator against the same frame on ONE context: the assembled framebuffers must be equal
        (the image does not depend on the partition).  A difference or an error moves the exchange to the fallback."""
        if self.comm is None:
            return
        ok, note = True, None
        try:
            self.init_params(W, H, spp, depth)
            self.frame(W, H)
            got = checksum_of(self.root)
            solo = self.srt.Renderer(self.devices[0])
            try:
                img = self.srt.render_image(self._scene, self._cam, W, H, spp, depth, renderer=solo)
                want = int(sum(int(p.astype("int64").sum()) for p in img["fb"]))
            finally:
                solo.close()
            if got != want:
                ok, note = False, "framebuffer checksum %d != single-context %d" % (got, want)
        except Exception as e:      # noqa: BLE001
            ok, note = False, "error %r" % (e,)
        if ok:
            self.gather_check = ("verified: %d-spp frame through the library's ncclGather (%d ranks) == the same frame rendered by one context "
                                 "(framebuffer checksum)" % (spp, self.world))
        else:
            log("library gather FAILED its verification frame (%s)" % note)
            self._fallback("library gather FAILED its verification frame: %s" % note)
            self.upload(self._scene, self._cam)
            self.gather_check = "failed: " + note

    def exchange_ms(self):
        return self.comm.last_gather_ms() if self.comm is not None else None

    def barrier(self):
        for d in self.devices:
            self.torch.cuda.synchronize(d)

    def reduce_sum(self, values):
        return list(values)

    def reduce_max(self, values):
        return list(values)

    def gather_rows(self, row):
        raise NotImplementedError      # per-rank rows are built from self.local directly

    def close(self):
        if self.comm is not None:
            self.comm.close()
        else:
            for r in self.local:
                r.close()


class OneProcessPerGpu:
    """torch.distributed.run: this process is ONE rank.  The exchange is the library's communicator (srt_comm_init_rank; the id
    travels through torch.distributed), verified against torch.distributed.gather before it carries a measurement; fallback:
    torch.distributed.gather."""
    launch_mode = "one process per GPU (torch.distributed.run, srt_comm_init_rank)"

    def __init__(self, srt, torch, dist, tiles, rank, world, local_rank, torch_gather=False, rehearse_gloo=False):
        self.srt, self.torch, self.dist, self.tiles = srt, torch, dist, tiles
        self.rank, self.world, self.rehearse = rank, world, rehearse_gloo
        self.red_dev = "cpu" if rehearse_gloo else "cuda"       # where the small statistics tensors of the reductions live
        self.r = srt.Renderer(local_rank)
        self.r.set_partition(rank, world)
        self.local, self.root = [self.r], (self.r if rank == 0 else None)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.comm, self.gather_check, self._local_tiles = None, None, None
        self.gather_via = "torch.distributed gather"
        if rehearse_gloo:
            self.launch_mode = "REHEARSAL: one process per rank on ONE GPU, torch.distributed on gloo (never used for reported numbers)"
        if torch_gather or rehearse_gloo:
            return
        # (1) a cheap LOCAL precheck agreed on by all ranks before anybody enters the collective ncclCommInitRank: a rank
        # that cannot load RCCL must not leave the others waiting in the bootstrap
        pre = torch.tensor([1.0 if srt.Comm.available() else 0.0], device=self.red_dev)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        if float(pre[0]) > 0.5:
            ident = [None]
            if rank == 0:
                try:
                    ident[0] = srt.Comm.unique_id()
                except Exception as e:      # noqa: BLE001 -- any failure here must not lose the measurement
                    log("library communicator unavailable (%r); using torch.distributed for the gather" % (e,))
            dist.broadcast_object_list(ident, src=0)      # every rank takes part, whatever rank 0 got
            if ident[0] is not None:
                try:
                    self.comm = srt.Comm.init_rank(self.r, ident[0], rank, world)
                except Exception as e:      # noqa: BLE001
                    log("rank %d: srt_comm_init_rank failed (%r); using torch.distributed for the gather" % (rank, e))
                    self.comm = None
            ok = torch.tensor([1.0 if self.comm is not None else 0.0], device=self.red_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok[0]) < 0.5 and self.comm is not None:      # every rank or none
                self.comm.close()
                self.comm = None
        else:
            log("rank %d: RCCL not loadable on every rank; using torch.distributed for the gather" % rank)
        if self.comm is None:
            self.gather_via = "torch.distributed gather (library communicator unavailable)"
        else:
            self.gather_via = "srt_render_frame_multi (ncclGather inside libsrt_hip.so, communicator from srt_comm_init_rank)"
            # every rank must gather the same number of planes (the library checks it once more on the first frame)
            planes = torch.tensor([3.0, -3.0], device=self.red_dev)
            dist.all_reduce(planes, op=dist.ReduceOp.MAX)
            assert float(planes[0]) == 3.0 and float(planes[1]) == -3.0

    def upload(self, scene, cam):
        self.r.upload_scene(scene); self.r.set_camera(cam); self.r.set_partition(self.rank, self.world)

    def init_params(self, W, H, spp, depth):
        self.r.init_device_params(W, H, spp, depth, 1984)

    def set_count(self, on):
        self.r.set_count_traversal(on)

    def _frame_torch_gather(self, W, H):
        torch, r = self.torch, self.r
        r.render_chunk(W, H, 0, 0, self.stream)
        _, n_floats, _, _ = r.tile_buffer()          # the exchange unit: the quantised framebuffer of this rank's tiles (12 B / pixel)
        if self._local_tiles is None or self._local_tiles.numel() != n_floats:
            self._local_tiles = torch.empty(n_floats, dtype=torch.float32, device="cuda")
        r.copy_tile_buffer(self._local_tiles.data_ptr(), self.stream)        # stream-ordered D2D into the tensor RCCL sends
        if self.rehearse:
            torch.cuda.current_stream().synchronize()
            gh = self.tiles.gather_tiles(self._local_tiles.cpu(), self.rank, self.world)
            g = gh.cuda() if self.rank == 0 else None
        else:
            g = self.tiles.gather_tiles(self._local_tiles, self.rank, self.world)      # the single collective of the path
        if self.rank == 0:
            r.scatter_tiles(g.data_ptr(), self.stream)
        torch.cuda.synchronize()          # (g is a temporary)

    def frame(self, W, H):
        if self.comm is not None:
            self.comm.render_frame(W, H, 0, 0)           # render + ONE ncclGather + scatter on rank 0, all enqueued by the library
            self.comm.synchronize()
        else:
            self._frame_torch_gather(W, H)

    def verify(self, W, H, depth, spp=8):
        """the library's RCCL gather against torch.distributed.gather on one cheap frame, before it carries the measurement: rank 0
        compares the assembled framebuffers; any difference or error falls back to torch.distributed.gather and the JSON says so."""
        if self.comm is None:
            return
        torch, dist = self.torch, self.dist
        verdict = 1.0
        try:
            self.init_params(W, H, spp, depth)
            self._frame_torch_gather(W, H)
            want = checksum_of(self.r) if self.rank == 0 else 0
            self.init_params(W, H, spp, depth)
            self.frame(W, H)
            got = checksum_of(self.r) if self.rank == 0 else 0
            if self.rank == 0 and got != want:
                verdict = 0.0
                log("library gather framebuffer checksum %d != torch gather %d" % (got, want))
        except Exception as e:      # noqa: BLE001
            verdict = 0.0
            log("rank %d: library gather failed in the verification frame (%r)" % (self.rank, e))
        v = torch.tensor([verdict], device=self.red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if float(v[0]) < 0.5:
            self.comm.close()
            self.comm = None
            self.gather_via = "torch.distributed gather (library gather FAILED its verification frame)"
            self.gather_check = "failed"
        else:
            self.gather_check = ("verified: %d-spp frame through the library's ncclGather == the same frame through torch.distributed.gather "
                                 "(framebuffer checksum on rank 0)" % spp)

    def exchange_ms(self):
        return self.comm.last_gather_ms() if self.comm is not None else None

    def barrier(self):
        self.dist.barrier()
        self.torch.cuda.synchronize()

    def _reduce(self, values, op):
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=op)
        return t.tolist()

    def reduce_sum(self, values):
        return self._reduce(values, self.dist.ReduceOp.SUM)

    def reduce_max(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX)

    def gather_rows(self, row):
        t = self.torch.zeros((self.world, len(row)), dtype=self.torch.float64, device=self.red_dev)
        for k, v in enumerate(row):
            t[self.rank, k] = float(v)
        self.dist.all_reduce(t)
        return t.tolist()

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.r.close()


def traversal_counts(job, W, H, depth):
    """V, T (node records / triangle tests per ray), rays per path and the share of NaN-direction rays from the instrumented kernel
    at 4 spp, all ranks, outside any timed region"""
    job.init_params(W, H, 4, depth)
    job.set_count(True)
    job.frame(W, H)
    loc = [0.0] * 8
    for r in job.local:
        st = r.stats()
        # util[4] / util[5]: lanes served by FRINGE / INNER steps = visits of FRINGE / INNER records
        for k, v in enumerate((st["rays"], st["node_visits"], st["tri_tests"], st["paths"], st["util"][2], st["util"][4], st["util"][5], st["hits"])):
            loc[k] += float(v)
    job.set_count(False)
    rays_c, V_c, T_c, paths_c, nan_c, vf_c, vi_c, hit_c = job.reduce_sum(loc)
    V, T = V_c / rays_c, T_c / rays_c
    # (util[4] is exact: a FRINGE step serves every lane that sits at a FRINGE record; util[5] is a utilisation figure: INNER visits = V - V_fringe)
    v_fringe, hits = vf_c / rays_c, hit_c / rays_c
    v_inner = V - v_fringe
    return dict(V=V, T=T, rays_per_path=rays_c / paths_c, nan_share=nan_c / rays_c, b_ray=V * NODE_BYTES + T * TRI_BYTES + MAT_BYTES,
                V_inner=v_inner, V_fringe=v_fringe, hits_per_ray=hits,
                b_ray_layout=v_inner * INNER_BYTES + v_fringe * FRINGE_BYTES + hits * SHADE_BYTES + MAT_BYTES)


def bytes_fields(tc):
    """the two per-ray byte figures of a record: SURVEY 8(d)'s formula and the same count in this build's record sizes"""
    return {"algorithmic_bytes_per_ray": tc["b_ray_layout"], "algorithmic_bytes_per_ray_survey_formula": tc["b_ray"],
            "algorithmic_bytes_note": "layout: V_inner x 64 + V_fringe x 96 + hits x 48 + 56 B (INNER / FRINGE / shading records of this build, 7 x 8 B of spectrum pairs per "
                                      "ray); survey formula: V x 64 + T x 48 + 56 with V = V_inner + V_fringe",
            "V_inner": tc["V_inner"], "V_fringe": tc["V_fringe"], "hits_per_ray": tc["hits_per_ray"]}


def timed_frames(job, W, H, spp, depth, steps, warmup, label=""):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones between two barriers; one step = seed the per-pixel RNG streams
    (init_device_params, rendering.cu:320-335), render, assemble the framebuffer on rank 0 -- every step produces the same image.
    Returns elapsed (max over ranks), total rays, and per-rank rows [kernel ms, rays per frame, exchange ms after own kernel]."""
    def step():
        job.init_params(W, H, spp, depth)
        job.frame(W, H)

    for k in range(warmup):
        t0 = time.time()
        step()
        job.barrier()
        if job.rank == 0:
            log("%swarmup %d/%d: %.2f s" % (label, k + 1, warmup, time.time() - t0))
    job.barrier()
    t0 = time.perf_counter()
    n_local = len(job.local)
    kernel_ms, rays, gather_ms = [[] for _ in range(n_local)], [0] * n_local, []
    for k in range(steps):
        ts = time.time()
        step()
        # per-step kernel time from the HIP events the library records on the launch stream around the render kernel alone
        for i, r in enumerate(job.local):
            kernel_ms[i].append(r.last_kernel_ms())
            rays[i] += r.stats()["rays"]
        g = job.exchange_ms()
        if g is not None:
            gather_ms.append(g)
        if job.rank == 0:
            log("%sstep %d/%d: kernel %.1f ms, wall %.2f s" % (label, k + 1, steps, max(x[-1] for x in kernel_ms), time.time() - ts))
    job.barrier()
    elapsed = time.perf_counter() - t0
    n = max(steps, 1)
    mean_k = [sum(x) / max(len(x), 1) for x in kernel_ms]
    mean_g = sum(gather_ms) / len(gather_ms) if gather_ms else 0.0
    total_rays = job.reduce_sum([float(sum(rays))])[0]
    elapsed = job.reduce_max([elapsed])[0]
    if n_local == job.world:          # this process sees every rank
        rows = [[mean_k[i], rays[i] / n, mean_g] for i in range(n_local)]
    else:
        rows = job.gather_rows([mean_k[0], rays[0] / n, mean_g])
    return dict(elapsed=elapsed, total_rays=total_rays, rows=rows, kms=max(x[0] for x in rows), rays_rank0=rows[0][1])


def lane_ops_entry(scene_id, tree=None):
    """the committed PMC entry of a scene: "scene_<id>", or -- when the run traverses another tree of the same scene (the builder's
    tree of a chain-bound launch next to the throughput-tuned one) -- the entry "scene_<id>_*" that was taken on that tree"""
    for f in LANE_OPS_FILES:
        try:
            doc = json.load(open(f))
        except Exception:      # noqa: BLE001
            continue
        e = doc.get("scene_%d" % scene_id)
        if tree is not None and (e is None or e.get("tree_sha256") != tree):
            for k, v in doc.items():
                if k.startswith("scene_%d_" % scene_id) and isinstance(v, dict) and v.get("tree_sha256") == tree:
                    e = v
                    break
        if e is not None:
            return e, os.path.relpath(f, ROOT)
    return None, None


def kernel_tie(renderer, entry, lib_path, tree=None):
    """which kernel variant the uploaded scene launches, the hash of its machine code in the LOADED library (tools/kernel_id.py: the
    kernel's bytes in the gfx950 code object; for PMC entries that predate it, the hash of the build's ISA listing), and how an
    imported PMC entry relates to it"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_id
    plan = renderer.launch_plan()
    nr, ac, pr = int(plan["narrow_refs"]), int(plan["all_cached"]), int(plan.get("paired", False))
    variant = kernel_id.variant_name(nr, ac, pr)
    code_hash, code_note = kernel_id.code_hash(lib_path, nr, ac, pr)
    isa_hash, isa_note = kernel_id.isa_hash(nr, ac, pr)
    hashes = {"code_sha256": code_hash, "isa_listing_sha256": isa_hash}
    tie = None
    if entry is not None:
        if entry.get("kernel_code_sha256") is not None:
            mine, theirs, what = code_hash, entry["kernel_code_sha256"], "machine code in the loaded library"
            note = code_note
        else:
            mine, theirs, what = isa_hash, entry.get("kernel_isa_sha256"), "ISA listing of the build"
            note = isa_note
        if mine is None:
            tie = "UNVERIFIED (%s)" % note
        elif theirs is None:
            tie = "UNVERIFIED (the imported PMC pass predates the hash tie)"
        elif theirs == mine and tree is not None and entry.get("tree_sha256") not in (None, tree):
            tie = "STALE (PMC pass was taken on this kernel binary but on another tree of the scene: %s, this run traverses %s)" % (entry["tree_sha256"][:16], tree[:16])
        elif theirs == mine:
            tie = "current (PMC pass taken on this kernel binary: %s sha256 %s%s)" % (what, mine[:16], ", and on this tree: sha256 %s" % tree[:16] if tree is not None and entry.get("tree_sha256") == tree else "")
        else:
            tie = "STALE (PMC pass was taken on kernel %s, this library is %s: %s)" % (theirs[:16], mine[:16], what)
    return plan, variant, hashes, tie


def traffic_of(entry, entry_file, pixels, rays, kms):
    """Fabric (HBM + Infinity Cache) bytes of ONE launch from the committed PMC passes of the scene: FETCH_SIZE / WRITE_SIZE taken in
    separate --pmc passes at two sample counts and split into a per-pixel part (RNG state, framebuffer: does not grow with spp) and a
    per-ray part (tree and shading records that miss L2) -- tools/pmc_to_lane_ops.py, which also applies the guide's gfx950 correction
    (FETCH_SIZE x 2 for 16-byte-per-lane loads) to the per-ray part.  Returns the fields a roofline record carries."""
    if entry is None or entry.get("fabric_bytes_per_ray") is None:
        return {"traffic": None}
    t = entry["fabric_bytes_per_pixel"] * pixels + entry["fabric_bytes_per_ray"] * rays
    return {"traffic": t, "traffic_GBs": t / (kms * 1e-3) / 1e9, "traffic_bytes_per_ray": t / max(rays, 1.0),
            "traffic_source": "IMPORTED from %s: %.1f B per pixel + %.2f B per ray (%s) x this launch's pixels and rays; not measured in this run"
                              % (entry_file, entry["fabric_bytes_per_pixel"], entry["fabric_bytes_per_ray"], entry.get("fabric_note", "FETCH_SIZE + WRITE_SIZE passes"))}


def small_roofline(renderer, scene_id, rays_per_launch, kms, lib_path, tree=None, pixels=0):
    """frac_arch of a secondary workload: imported lane-ops per ray of ITS kernel variant / scene (hash-tied) x live rays / live kernel time"""
    entry, entry_file = lane_ops_entry(scene_id, tree)
    plan, variant, hashes, tie = kernel_tie(renderer, entry, lib_path, tree)
    out = {"kernel": variant, "tree_sha256": tree, "kernel_code_sha256": hashes["code_sha256"], "kernel_isa_sha256": hashes["isa_listing_sha256"], "frac_arch": None, "achieved_source": None}
    if entry is not None:
        ach = rays_per_launch * entry["lane_ops_per_ray"] / (kms * 1e-3) / 1e9
        out.update(achieved=ach, unit="G lane-op/s", frac_arch=ach / ARCH_PEAK_GLANEOPS, lanes_per_valu_instruction=entry.get("lanes_per_valu_instruction"),
                   achieved_source="%s: %.1f useful VALU lane-ops per ray IMPORTED from %s (kernel %s) x live rays / live kernel time" %
                                   (tie, entry["lane_ops_per_ray"], entry_file, entry.get("kernel", "?")))
        out.update(traffic_of(entry, entry_file, pixels, rays_per_launch, kms))
    else:
        out["achieved_source"] = "no PMC pass of scene %d under profiles/" % scene_id
    return out


PROFILE_ORDER = True      # --no-profile-order clears it


def tree_sha256(scene):
    """identity of the tree a scene is traversed with (topology, child order, boxes): an imported per-ray counter figure belongs to
    the kernel AND the tree it was taken on"""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for a in scene.bvh():
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def build_scene(srt, job, scene_id, bvh, W, H, depth, tune=None):
    """The scene as every frame traverses it, built outside any timed region (the reference builds its tree in create_bvh_kernel before
    the render, scene/scene.cu:9-20).  For this build's own SAH trees, when the launch is throughput-bound (at least 6 pixels per
    persistent lane of a rank's launch): topology optimisation + child order from a probe frame (srt.tune_tree_for_throughput).
    Launches with fewer pixels per lane are bound by their longest pixel chain, which a tree with less TOTAL work does not shorten
    (measured slower), and keep the builder's tree.  Deterministic: every rank arrives at the same tree.  Returns the scene, a
    description, and {tuned, tree_build_s, tree_tuning_s} (wall seconds of the builder and of the tuning calls incl. their probe frames)."""
    if tune is None:
        tune = PROFILE_ORDER
    t0 = time.time()
    scene = srt.Scene.builtin(scene_id, 0).build_bvh(bvh, 1984)
    info = {"tuned": False, "tree_build_s": time.time() - t0, "tree_tuning_s": 0.0}
    if bvh != 1:
        return scene, "the reference builder's tree", info
    r = job.local[0]
    r.upload_scene(scene)
    ppl = srt.pixels_per_lane(r, W, H, job.world)
    if not tune:
        return scene, "SAH builder's tree, nearer child to the camera first (--no-profile-order)" if not PROFILE_ORDER else "SAH builder's tree, nearer child to the camera first (untuned, for comparison)", info
    if ppl < 6.0:
        return scene, "SAH builder's tree, nearer child to the camera first (%.1f pixels per lane: chain-bound launch, no throughput tuning)" % ppl, info
    t0 = time.time()
    note = srt.tune_tree_for_throughput(r, scene, W, H, depth)
    info.update(tuned=True, tree_tuning_s=time.time() - t0)
    return scene, "SAH builder's tree tuned for throughput (%.1f pixels per lane): %s" % (ppl, note), info


def secondary_workload(srt, job, scene_id, bvh, W, H, spp, depth, with_roofline=True):
    """one frame of another BASELINE configuration on the same ranks, outside the headline's timed region"""
    t_build = time.time()
    ok, err = 1.0, None
    try:
        scene, order_note, tree_info = build_scene(srt, job, scene_id, bvh, W, H, depth)
        cam = scene.default_camera(W, H)
        job.upload(scene, cam)
    except Exception as e:      # noqa: BLE001
        ok, err = 0.0, e
    # every rank or none enters the collectives below: a rank that could not set the workload up must not leave the others waiting
    if min(job.reduce_max([-ok])) > -0.5 or ok < 0.5:
        raise RuntimeError("secondary workload not set up on every rank (this rank: %r)" % (err,))
    t_build = time.time() - t_build
    tc = traversal_counts(job, W, H, depth)
    tf = timed_frames(job, W, H, spp, depth, 1, 0, label="[%s %dx%d %d spp] " % (SCENE_NAMES.get(scene_id, scene_id), W, H, spp))
    rec = None
    if job.rank == 0:
        rec = {"workload": "%s (%d tris, %d BVH nodes), %dx%d, %d spp, depth %d" % (SCENE_NAMES.get(scene_id, "scene %d" % scene_id), scene.n_tris, scene.n_nodes, W, H, spp, depth),
               "scene_id": scene_id, "bvh": "SAH" if bvh == 1 else "reference builder", "child_order": order_note,
               "value": tf["total_rays"] / tf["elapsed"] / 1e6, "unit": "Mray/s", "frames": 1, "ms": tf["elapsed"] * 1e3, "kernel_ms": tf["kms"],
               "rays": tf["total_rays"], "V": tc["V"], "T": tc["T"], "rays_per_path": tc["rays_per_path"],
               "nan_direction_rays_pct": 100.0 * tc["nan_share"],
               "fb_checksum": checksum_of(job.root), "scene_build_s": t_build, "tree_build_s": tree_info["tree_build_s"],
               "tree_tuning_s": tree_info["tree_tuning_s"], "tree_sha256": tree_sha256(scene)}
        rec.update(bytes_fields(tc))
        if job.world > 1:
            rec["per_rank_kernel_ms"] = [round(x[0], 3) for x in tf["rows"]]
            rec["per_rank_rays"] = [int(x[1]) for x in tf["rows"]]
        if with_roofline and job.world == 1:
            rec["roofline"] = small_roofline(job.root, scene_id, tf["rays_rank0"], tf["kms"], srt.binding.LIB_PATH, tree_sha256(scene), pixels=W * H)
            rec["roofline"]["hbm_algorithmic_GBs"] = tf["rays_rank0"] * tc["b_ray_layout"] / (tf["kms"] * 1e-3) / 1e9
            rec["frac_arch"] = rec["roofline"]["frac_arch"]
            rec["achieved_source"] = rec["roofline"]["achieved_source"]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", type=int, default=100)          # SRT_SCENE_RANDOM_SPHERES
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--bvh", type=int, default=1)              # SRT_BVH_SAH for the synthetic scenes
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the issue-rate microkernel and the copy-rate measurement (profiling passes)")
    ap.add_argument("--no-profile-order", action="store_true", help="keep the SAH builder's child order (nearer child to the camera first) instead of the profiled one")
    ap.add_argument("--no-other-configs", action="store_true", help="skip cfg 2 / cfg 4 / cfg 5's scene at 512 spp (N = 1 block `other_configs`)")
    ap.add_argument("--no-builders-tree", action="store_true", help="skip the comparison frames on the SAH builder's untuned tree (`value_builders_tree`)")
    ap.add_argument("--time-budget", type=float, default=420.0, help="seconds since process start after which the cfg 5 sub-record is not started (the driver allows 600 s per invocation)")
    ap.add_argument("--cfg5-spp", type=int, default=4096, help="samples of the cfg 5 sub-record (BASELINE: 4096); 0 = skip it")
    ap.add_argument("--cfg5-size", default="3840x2160", help="image size of the cfg 5 sub-record (BASELINE: 3840x2160)")
    ap.add_argument("--torch-gather", action="store_true", help="one process per GPU: gather through torch.distributed instead of the library's RCCL communicator")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the process-per-GPU control flow on ONE GPU: every rank uses cuda:0, torch.distributed runs on gloo and the gather goes through host memory (never used for reported numbers)")
    args = ap.parse_args()
    global PROFILE_ORDER
    PROFILE_ORDER = not args.no_profile_order

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launcher = "WORLD_SIZE" in os.environ and world > 1
    if launcher and args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    n_dev = torch.cuda.device_count()

    import __graft_entry__
    srt = __graft_entry__._pkg()

    # ---- the ranks --------------------------------------------------------------------------------------------------------
    if launcher:
        import torch.distributed as dist
        from importlib import import_module
        tiles = import_module("cuda-spectral-ray-tracer_amd.tiles")
        if args.rehearse_gloo or local_rank >= n_dev:      # (launcher restricted each rank's visible devices to its own GPU)
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        job = OneProcessPerGpu(srt, torch, dist, tiles, rank, world, local_rank, args.torch_gather, args.rehearse_gloo)
    elif args.gpus > 1:
        # no launcher: this process drives all N GPUs.  (SRT_COMM_TEST_SAME_DEVICE=1 with the test transport of tests/cpp/mock_rccl.cpp
        # puts the N ranks on device 0: the single-GPU rehearsal of this path, tests/test_gpu_parity.py)
        same = os.environ.get("SRT_COMM_TEST_SAME_DEVICE") == "1" and os.environ.get("SRT_RCCL_LIB")
        if not same and args.gpus > n_dev:
            raise SystemExit("bench.py --gpus %d: only %d GPUs are visible to this process" % (args.gpus, n_dev))
        torch.cuda.set_device(0)
        world = args.gpus
        job = OneProcessManyGpus(srt, torch, [0] * world if same else list(range(world)))
        if same:
            job.launch_mode += " -- REHEARSAL: all ranks on device 0 over the test transport (never used for reported numbers)"
    else:
        torch.cuda.set_device(local_rank if local_rank < n_dev else 0)
        job = OneGpu(srt, torch, torch.cuda.current_device())

    # ---- inputs, resident in HBM before the timed region ------------------------------------------------
    W, H = args.width, args.height
    scene, order_note, tree_info = build_scene(srt, job, args.scene, args.bvh, W, H, args.depth)
    scene_tree = tree_sha256(scene)
    cam = scene.default_camera(W, H)
    job.upload(scene, cam)

    # ---- V, T from the instrumented kernel, outside the timed region
    tc = traversal_counts(job, W, H, args.depth)

    # ---- calibration of the roofs, outside the timed region (rank 0, N = 1) -------------------------------------------
    calib, copy_gbs = None, None
    if job.world == 1 and not args.no_calibration:
        calib = job.root.calibrate(0, 4, 40000)            # v_add_f32, 4 waves / SIMD, one workgroup per CU
        copy_gbs = measure_hbm_copy_gbs(torch)
        log("calibration: %.1f G wave-instr/s (%.3f / cycle / SIMD at %.2f GHz; wave Mcycles min/mean/max %.2f/%.2f/%.2f); copy %.0f GB/s" %
            (calib["instr_per_s"] / 1e9, calib["instr_per_cycle_per_simd"], calib["clock_ghz"], calib["wave_cycles_min"] / 1e6,
             calib["wave_cycles_mean"] / 1e6, calib["wave_cycles_max"] / 1e6, copy_gbs))

    # ---- the exchange path is checked on one cheap frame before it carries a measurement (RCCL has carried this path on mock
    # transports and one-rank worlds only: no multi-GPU hardware has run it before the first driver run that has such a node)
    if job.world > 1:
        job.verify(W, H, args.depth)

    # ---- the timed region --------------------------------------------------------------------------------------------------
    tf = timed_frames(job, W, H, args.spp, args.depth, args.steps, args.warmup)
    elapsed, total_rays, kms = tf["elapsed"], tf["total_rays"], tf["kms"]

    out = None
    if job.rank == 0:
        steps = max(args.steps, 1)
        headline = job.world == 1 and (args.scene, W, H, args.spp, args.depth, args.bvh) == (100, 1920, 1080, 1024, 16, 1)
        mray = total_rays / elapsed / 1e6
        rays_per_launch_rank0 = tf["rays_rank0"]
        b_ray = tc["b_ray_layout"]
        # ---- roofline of the dominant kernel (render_kernel) on this rank -------------------------------------------
        entry, entry_file = lane_ops_entry(args.scene, scene_tree)
        plan, variant, hashes, tie = kernel_tie(job.root, entry, srt.binding.LIB_PATH, scene_tree)
        roof = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "G lane-op/s", "frac": None, "frac_arch": None, "traffic": None,
                "peak_arch": ARCH_PEAK_GLANEOPS,
                "peak_arch_source": "256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles; 157.3 TFLOP/s fp32 FMA / 2)"}
        roof["kernel_code_sha256"], roof["kernel_isa_sha256"] = hashes["code_sha256"], hashes["isa_listing_sha256"]
        roof["tree_sha256"] = scene_tree
        roof["kernel"] = variant + ", %d waves per CU, %d of the inner records in LDS" % (plan["waves_per_cu"], plan["n_cached"])
        roof["library"] = os.path.relpath(srt.binding.LIB_PATH, ROOT)
        if entry is not None:
            lane_ops_per_ray = entry["lane_ops_per_ray"]
            roof["achieved"] = rays_per_launch_rank0 * lane_ops_per_ray / (kms * 1e-3) / 1e9
            roof["achieved_source"] = ("%s: useful VALU lane-ops per ray = %.1f (SQ_THREAD_CYCLES_VALU / rays, rocprofv3 --pmc pass of kernel %s, "
                                       "IMPORTED from %s) x %.4g rays per launch / %.2f ms render-kernel time, both measured in this run"
                                       % (tie, lane_ops_per_ray, entry.get("kernel", "?"), entry_file, rays_per_launch_rank0, kms))
            roof["lanes_per_valu_instruction"] = entry.get("lanes_per_valu_instruction")
            roof["wave_time_split"] = entry.get("wave_time_split")
            if entry.get("fabric_bytes_per_ray") is not None:
                roof.update(traffic_of(entry, entry_file, W * H, rays_per_launch_rank0, kms))
            elif entry.get("hbm_bytes_per_launch_1024spp") is not None and headline:      # (entries of rounds 2-4)
                roof["traffic"] = entry["hbm_bytes_per_launch_1024spp"]
                roof["traffic_source"] = "IMPORTED from %s (FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this workload), not measured in this run" % entry_file
        if calib is not None:
            roof["peak"] = calib["instr_per_s"] * 64 / 1e9
            roof["peak_source"] = ("measured in this run: srt_calibrate kind 0 (independent v_add_f32, 4 waves/SIMD, one workgroup on each of %d CUs): "
                                   "%.1f G wave-instr/s by the wall clock = %.3f per cycle per SIMD at %.2f GHz (all waves' instructions / cycles of the "
                                   "LAST wave of a SIMD; architectural 0.5) x 64 lanes" % (calib["n_cu"], calib["instr_per_s"] / 1e9,
                                                                                        calib["instr_per_cycle_per_simd"], calib["clock_ghz"]))
        if roof["achieved"] is not None:
            roof["frac_arch"] = roof["achieved"] / ARCH_PEAK_GLANEOPS
            if roof["peak"]:
                roof["frac"] = roof["achieved"] / roof["peak"]
        hbm_alg = rays_per_launch_rank0 * b_ray / (kms * 1e-3) / 1e9
        roof["hbm"] = {"algorithmic_GBs": hbm_alg, "algorithmic_bytes_per_ray": b_ray, "algorithmic_bytes_per_ray_survey_formula": tc["b_ray"],
                       "measured_copy_peak_GBs": copy_gbs, "spec_peak_GBs": HBM_SPEC_GBS,
                       "frac_of_spec_peak": hbm_alg / HBM_SPEC_GBS,
                       "frac_of_measured_peak": (hbm_alg / copy_gbs) if copy_gbs else None,
                       "note": "V_inner*64 + V_fringe*96 + hits*48 + 56 bytes per ray (this build's record sizes; SURVEY 8(d)'s V*64 + T*48 + 56 beside it) x rays per launch / kernel time.  A value above 1 means these bytes "
                               "never reach HBM: the inner tree is LDS resident and the rest is L2 resident (measured HBM traffic: roofline.traffic) -- "
                               "HBM is not the roof of this kernel, the vector issue port is"}
        t_d2h = time.perf_counter()
        checksum = checksum_of(job.root)       # outside the timed region: image checksum, identical for every N
        d2h_ms = (time.perf_counter() - t_d2h) * 1e3
        out = {
            "metric": "Mray/s", "value": mray, "unit": "Mray/s", "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (%d tris, %d BVH nodes), %dx%d, %d spp, depth %d" %
                                   (SCENE_NAMES.get(args.scene, "scene %d" % args.scene), scene.n_tris, scene.n_nodes, W, H, args.spp, args.depth),
                       "scene_id": args.scene,
                       "bvh": BVH_NAMES.get(args.bvh, str(args.bvh)), "child_order": order_note, "tree_sha256": scene_tree,
                       "tree_build_s": tree_info["tree_build_s"], "tree_tuning_s": tree_info["tree_tuning_s"],
                       "nan_direction_rays": "%.2f %% of the counted rays have a NaN direction (Sellmeier quirk Q1) and are answered 'miss' without walking the tree" % (100.0 * tc["nan_share"]),
                       "tiles": "8x8 px per wave, rank = tile % n_gpus", "gather": job.gather_via, "launch_mode": job.launch_mode},
            "mpath_per_s": (W * H * args.spp * steps) / elapsed / 1e6,
            "rays_per_path": tc["rays_per_path"], "node_records_per_ray_V": tc["V"], "tri_tests_per_ray_T": tc["T"],
            "kernel_ms_per_step": kms, "fb_checksum": checksum,
            # SURVEY 8(d)'s wall clock includes the hand-over of the framebuffer to the host; `value` / `ms_per_step` end with the
            # image in HBM (the boundary returns device memory), this adds the synchronous D2H of the three quantised planes
            "d2h_ms": d2h_ms, "ms_per_step_incl_d2h": elapsed / steps * 1e3 + d2h_ms,
            "mray_per_s_incl_d2h": total_rays / steps / (elapsed / steps + d2h_ms * 1e-3) / 1e6,
            "roofline": roof,
        }
        out.update(bytes_fields(tc))
        if job.world > 1:
            rows = tf["rows"]
            out["per_rank"] = {"kernel_ms": [round(x[0], 3) for x in rows], "rays_per_frame": [int(x[1]) for x in rows],
                               "exchange_ms_after_own_kernel": [round(x[2], 3) for x in rows] if "ncclGather" in job.gather_via else None}
            out["gather_check"] = job.gather_check

    # ---- the same workload on the BUILDER's tree (no tuning): keeps the scene-side gain apart from the kernel-side one, round over round.
    # Every rank takes part (collectives inside); outside the timed region; the tuned scene is uploaded again afterwards.
    builders = None
    if tree_info["tuned"] and not args.no_builders_tree:
        try:
            scene_b, note_b, _ = build_scene(srt, job, args.scene, args.bvh, W, H, args.depth, tune=False)
            job.upload(scene_b, cam)
            tc_b = traversal_counts(job, W, H, args.depth)
            tf_b = timed_frames(job, W, H, args.spp, args.depth, 2, 1, label="[builder's tree] ")
            if job.rank == 0:
                builders = {"value": tf_b["total_rays"] / tf_b["elapsed"] / 1e6, "unit": "Mray/s", "ms_per_step": tf_b["elapsed"] / 2 * 1e3, "kernel_ms_per_step": tf_b["kms"],
                            "steps": 2, "warmup": 1, "child_order": note_b, "tree_sha256": tree_sha256(scene_b), "V": tc_b["V"], "T": tc_b["T"],
                            "V_inner": tc_b["V_inner"], "V_fringe": tc_b["V_fringe"], "fb_checksum": checksum_of(job.root)}
        except Exception as e:      # noqa: BLE001
            log("builder's-tree frame failed: %r" % (e,))
            builders = {"value": None, "error": repr(e)}
        job.upload(scene, cam)
    if out is not None:
        if builders is not None:
            out["value_builders_tree"] = builders.get("value")
            out["builders_tree"] = builders
        elif not tree_info["tuned"]:
            out["value_builders_tree"] = out["value"]
            out["builders_tree"] = {"note": "the headline was measured on the builder's tree itself (no throughput tuning for this launch)", "tree_sha256": scene_tree}

    # ---- the CPU baseline (N = 1 only), while the headline scene is still uploaded -------------------------------------------
    if out is not None and not args.no_cpu_baseline and job.world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(srt, scene, cam, W, H, args.depth, args.bvh, gpu_renderer=job.root)
            # the extrapolated wall time of ONE frame of this workload on those host cores (the rate is spp independent)
            if out["cpu_baseline"].get("value"):
                out["cpu_baseline"]["extrapolated_wall_s_per_frame"] = (total_rays / max(args.steps, 1)) / (out["cpu_baseline"]["value"] * 1e6)
                out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:   # noqa: BLE001 -- the checker is optional for the measurement itself
            out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": usable_cores()[0], "kind": "port", "sample": "failed: %r" % (e,)}
        try:
            out["cpu_baseline"]["cfg1_single_thread"] = cfg1_single_thread(srt)
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"]["cfg1_single_thread"] = {"value": None, "sample": "failed: %r" % (e,)}

    # ---- the other BASELINE configurations, one frame each, after the timed region (every rank takes part: collectives inside) ----
    others, cfg5 = [], None
    if job.world == 1 and not args.no_other_configs:
        for sid, bvh, w, h, spp, depth, name in ((100, 1, 1280, 720, 256, 16, "cfg 2"), (1, 0, 1920, 1080, 2048, 16, "cfg 4"), (101, 1, 3840, 2160, 512, 16, "cfg 5's scene at 512 spp")):
            try:
                rec = secondary_workload(srt, job, sid, bvh, w, h, spp, depth)
                rec["baseline_config"] = name
            except Exception as e:      # noqa: BLE001
                rec = {"baseline_config": name, "value": None, "error": repr(e)}
            others.append(rec)
    if args.cfg5_spp > 0:
        cw, ch = [int(x) for x in args.cfg5_size.lower().split("x")]
        # (estimated cost: 13 s per 3840x2160 x 4096 spp frame on one GPU, shared by the ranks, x 1.5 + set-up; every rank takes the same
        # decision: the clock of rank 0 is what counts)
        est = 13.0 * (cw * ch / (3840.0 * 2160.0)) * (args.cfg5_spp / 4096.0) / max(job.world, 1) * 1.5 + 8.0
        spent = job.reduce_max([time.time() - T_PROCESS_START])[0]
        if spent + est > args.time_budget:
            log("cfg 5 sub-record skipped: %.0f s spent + %.0f s estimated > --time-budget %.0f s" % (spent, est, args.time_budget))
            cfg5 = {"value": None, "skipped": "time budget: %.0f s since process start + %.0f s estimated > %.0f s" % (spent, est, args.time_budget)}
            args.cfg5_spp = 0
    if args.cfg5_spp > 0:
        try:
            cfg5 = secondary_workload(srt, job, 101, 1, cw, ch, args.cfg5_spp, 16)
            if cfg5 is not None:
                cfg5["baseline_config"] = "cfg 5 (BASELINE: 3840x2160, 4096 spp, tiles across the GPUs of the node, one RCCL gather)"
        except Exception as e:      # noqa: BLE001
            log("cfg 5 sub-record failed: %r" % (e,))
            cfg5 = {"value": None, "error": repr(e)}
    if out is not None:
        if others:
            out["other_configs"] = others
        if cfg5 is not None:
            out["cfg5"] = cfg5
        print(json.dumps(out), flush=True)

    job.close()
    if launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
