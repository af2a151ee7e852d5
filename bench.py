#!/usr/bin/env python3
"""bench.py -- headline benchmark of the spectral path-tracing hot path on MI355X.

Metric (BASELINE.json): Mray/s on the synthetic "random spheres" scene, 1920x1080, 1024 spp, depth 16
(config[2]; a ray = one closest-hit query).  One *step* = one full render of that image: every rank
renders its interleaved 8x8 tiles with the HIP kernel, ONE gather (RCCL over xGMI, issued by the library:
srt_render_frame_multi) brings the compact tile buffers to rank 0, which scatters them into the
block-linear planar framebuffer in HBM.  Scene, camera and RNG states are resident in HBM before the timed
region.  The image is fixed, so N GPUs split the same work: scaling is "strong".

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.

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
binary by a hash of its ISA listing: a mismatch is reported as "achieved_source": "STALE ...") x rays per launch (live) /
render-kernel time (live, HIP events on the launch stream).

`cpu_baseline` is the CPU oracle (a port: the reference has no CPU path) timed on this box's USABLE host cores
(scheduler affinity and cgroup quota, not os.cpu_count()) on a bounded sample of the same workload, its one-thread rate
on the same scene, and BASELINE.md's "config 1, one thread, full size" leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_SPEC_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy rate is measured below
NODE_BYTES, TRI_BYTES, MAT_BYTES = 64, 48, 56
LANE_OPS_FILES = [os.path.join(ROOT, "profiles", r, "lane_ops_per_ray.json") for r in ("r03", "r02")]      # newest first
ARCH_PEAK_GLANEOPS = 256 * 4 * 32 * 2.4      # G lane-op/s: 256 CU x 4 SIMD x 32 lanes per clock x 2.4 GHz (157.3 TFLOP/s fp32 FMA / 2)
SCENE_NAMES = {0: "reference CORNELL scene", 1: "reference PRISM scene", 2: "reference TRIS scene", 100: "random-spheres tri scene",
               101: "100k-triangle mesh in the Cornell shell"}


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
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: gather through torch.distributed instead of the library's RCCL communicator")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 control flow on ONE GPU: every rank uses cuda:0, torch.distributed runs on gloo and the gather goes through host memory (never used for reported numbers)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_gloo or local_rank >= torch.cuda.device_count():      # (launcher restricted each rank's visible devices to its own GPU)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if args.rehearse_gloo else "cuda"       # where the small statistics tensors of the reductions live

    import __graft_entry__
    srt = __graft_entry__._pkg()
    from importlib import import_module
    tiles = import_module("cuda-spectral-ray-tracer_amd.tiles")

    # ---- inputs, resident in HBM before the timed region ------------------------------------------------
    scene = srt.Scene.builtin(args.scene, 0).build_bvh(args.bvh, 1984)
    W, H = args.width, args.height
    cam = scene.default_camera(W, H)
    r = srt.Renderer(local_rank)
    r.upload_scene(scene)
    r.set_camera(cam)
    r.set_partition(rank, world)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- the exchange path: the library's own RCCL communicator (srt_comm_*, behind the C-ABI); torch.distributed only
    # carries the 128-byte communicator id, the barriers and the statistics.  If the communicator cannot be formed the
    # gather falls back to torch.distributed (same bytes, same xGMI links) and the JSON says so.
    comm, gather_via, gather_check = None, "none (1 GPU)", None
    if world > 1:
        gather_via = "torch.distributed gather"
        if not args.torch_gather and not args.rehearse_gloo:
            # (1) a cheap LOCAL precheck agreed on by all ranks before anybody enters the collective ncclCommInitRank: a rank
            # that cannot load RCCL must not leave the others waiting in the bootstrap
            pre = torch.tensor([1.0 if srt.Comm.available() else 0.0], device=red_dev)
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
                        comm = srt.Comm.init_rank(r, ident[0], rank, world)
                        gather_via = "srt_render_frame_multi (ncclGather inside libsrt_hip.so)"
                    except Exception as e:      # noqa: BLE001
                        log("rank %d: srt_comm_init_rank failed (%r); using torch.distributed for the gather" % (rank, e))
                        comm = None
                ok = torch.tensor([1.0 if comm is not None else 0.0], device=red_dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok[0]) < 0.5 and comm is not None:      # every rank or none
                    comm.close()
                    comm = None
            else:
                log("rank %d: RCCL not loadable on every rank; using torch.distributed for the gather" % rank)
            if comm is None:
                gather_via = "torch.distributed gather (library communicator unavailable)"

    # ---- V, T (node records / triangle tests per ray) from the instrumented kernel, outside the timed region
    r.init_device_params(W, H, 4, args.depth, 1984)
    r.set_count_traversal(True)
    r.render_chunk(W, H, 0, 0, stream)
    st = r.stats()
    cnt = torch.tensor([st["rays"], st["node_visits"], st["tri_tests"], st["paths"], st["util"][2]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(cnt)
    rays_c, V_c, T_c, paths_c, nan_c = [float(x) for x in cnt.tolist()]
    V, T, rays_per_path = V_c / rays_c, T_c / rays_c, rays_c / paths_c
    b_ray = V * NODE_BYTES + T * TRI_BYTES + MAT_BYTES
    r.set_count_traversal(False)

    # ---- calibration of the roofs, outside the timed region (rank 0, N = 1) -------------------------------------------
    calib, copy_gbs = None, None
    if rank == 0 and world == 1 and not args.no_calibration:
        calib = r.calibrate(0, 4, 40000)            # v_add_f32, 4 waves / SIMD, one workgroup per CU
        copy_gbs = measure_hbm_copy_gbs(torch)
        log("calibration: %.1f G wave-instr/s (%.3f / cycle / SIMD at %.2f GHz; wave Mcycles min/mean/max %.2f/%.2f/%.2f); copy %.0f GB/s" %
            (calib["instr_per_s"] / 1e9, calib["instr_per_cycle_per_simd"], calib["clock_ghz"], calib["wave_cycles_min"] / 1e6,
             calib["wave_cycles_mean"] / 1e6, calib["wave_cycles_max"] / 1e6, copy_gbs))

    r.init_device_params(W, H, args.spp, args.depth, 1984)

    local_tiles = None      # torch-owned staging tensor for the torch.distributed gather

    def frame_torch_gather():
        nonlocal local_tiles
        r.render_chunk(W, H, 0, 0, stream)
        _, n_floats, _, _ = r.tile_buffer()          # the exchange unit: the quantised framebuffer of this rank's tiles (12 B / pixel)
        if local_tiles is None or local_tiles.numel() != n_floats:
            local_tiles = torch.empty(n_floats, dtype=torch.float32, device="cuda")
        r.copy_tile_buffer(local_tiles.data_ptr(), stream)        # stream-ordered D2D into the tensor RCCL sends
        if args.rehearse_gloo:
            torch.cuda.current_stream().synchronize()
            gh = tiles.gather_tiles(local_tiles.cpu(), rank, world)
            g = gh.cuda() if rank == 0 else None
        else:
            g = tiles.gather_tiles(local_tiles, rank, world)      # the single collective of the path
        if rank == 0:
            r.scatter_tiles(g.data_ptr(), stream)
            if args.rehearse_gloo:
                torch.cuda.current_stream().synchronize()          # g is a temporary

    def frame_library_gather():
        comm.render_frame(W, H, 0, 0)           # render + ONE ncclGather + scatter on rank 0, all enqueued by the library
        comm.synchronize()

    def checksum_rank0():
        if rank != 0:
            return 0
        fb = r.read_fb()
        return int(sum(int(p.astype("int64").sum()) for p in fb))

    # (2) the library's RCCL gather has been exercised on ONE GPU only (world 1 returns before the gather): before it carries the
    # measurement, one cheap frame (8 spp) goes through BOTH exchange paths and rank 0 compares the assembled framebuffers; any
    # difference or error falls back to the torch.distributed gather (validated on hardware in round 1) and the JSON says so.
    if comm is not None:
        verdict = 1.0
        try:
            r.init_device_params(W, H, 8, args.depth, 1984)
            frame_torch_gather(); torch.cuda.synchronize()
            want = checksum_rank0()
            r.init_device_params(W, H, 8, args.depth, 1984)
            frame_library_gather()
            got = checksum_rank0()
            if rank == 0 and got != want:
                verdict = 0.0
                log("library gather framebuffer checksum %d != torch gather %d" % (got, want))
        except Exception as e:      # noqa: BLE001
            verdict = 0.0
            log("rank %d: library gather failed in the verification frame (%r)" % (rank, e))
        v = torch.tensor([verdict], device=red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if float(v[0]) < 0.5:
            comm.close()
            comm = None
            gather_via = "torch.distributed gather (library gather FAILED its verification frame)"
            gather_check = "failed"
        else:
            gather_check = "verified: 8-spp frame through the library's ncclGather == the same frame through torch.distributed.gather (framebuffer checksum on rank 0)"

    def step():
        # one step = one complete frame: seed the per-pixel RNG streams (init_device_params, rendering.cu:320-335), render,
        # assemble the framebuffer on rank 0.  Every step therefore produces the same image (fb_checksum).
        r.init_device_params(W, H, args.spp, args.depth, 1984)
        if world == 1:
            r.render_chunk(W, H, 0, 0, stream)
            r.scatter_tiles(None, stream)
        elif comm is not None:
            frame_library_gather()
        else:
            frame_torch_gather()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        t0 = time.time()
        step()
        barrier()
        if rank == 0:
            log("warmup %d/%d: %.2f s" % (k + 1, args.warmup, time.time() - t0))

    barrier()
    t0 = time.perf_counter()
    kernel_ms, gather_ms, rays_local = [], [], 0
    for k in range(args.steps):
        ts = time.time()
        step()
        # per-step kernel time from the HIP events the library records on the launch stream; reading it waits for the
        # kernel only, and is inside the timed region on purpose (it costs one event sync).
        kernel_ms.append(r.last_kernel_ms())
        if comm is not None:
            gather_ms.append(comm.last_gather_ms())
        rays_local += r.stats()["rays"]
        if rank == 0:
            log("step %d/%d: kernel %.1f ms, wall %.2f s" % (k + 1, args.steps, kernel_ms[-1], time.time() - ts))
    barrier()
    elapsed = time.perf_counter() - t0

    my_kms = sum(kernel_ms) / max(len(kernel_ms), 1)
    my_gms = sum(gather_ms) / max(len(gather_ms), 1) if gather_ms else 0.0
    tot = torch.tensor([float(rays_local), elapsed, my_kms], dtype=torch.float64, device=red_dev)
    per_rank = None
    if world > 1:
        rays_t = tot[0:1].clone(); dist.all_reduce(rays_t)
        mx = tot[1:3].clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        total_rays, elapsed, kms = float(rays_t[0]), float(mx[0]), float(mx[1])
        # per-rank diagnosis of the first real multi-GPU lines: render-kernel ms, rays, and the time between the end of the rank's
        # kernel and the end of the exchange (gather + waiting for the slowest rank [+ scatter on rank 0])
        mine = torch.zeros((world, 3), dtype=torch.float64, device=red_dev)
        mine[rank, 0], mine[rank, 1], mine[rank, 2] = my_kms, float(rays_local) / max(args.steps, 1), my_gms
        dist.all_reduce(mine)
        per_rank = {"kernel_ms": [round(float(x), 3) for x in mine[:, 0].tolist()], "rays_per_frame": [int(x) for x in mine[:, 1].tolist()],
                    "exchange_ms_after_own_kernel": [round(float(x), 3) for x in mine[:, 2].tolist()] if comm is not None else None}
    else:
        total_rays, elapsed, kms = float(tot[0]), float(tot[1]), float(tot[2])

    if rank == 0:
        steps = max(args.steps, 1)
        headline = world == 1 and (args.scene, W, H, args.spp, args.depth, args.bvh) == (100, 1920, 1080, 1024, 16, 1)
        mray = total_rays / elapsed / 1e6
        rays_per_launch_rank0 = rays_local / steps
        # ---- roofline of the dominant kernel (render_kernel) on this rank -------------------------------------------
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_id
        plan = r.launch_plan()
        variant = "render_kernel<0,%d,%d>" % (int(plan["narrow_refs"]), int(plan["all_cached"]))
        my_hash, hash_note = kernel_id.isa_hash(int(plan["narrow_refs"]), int(plan["all_cached"]))
        entry, entry_file = None, None
        for f in LANE_OPS_FILES:
            try:
                e = json.load(open(f)).get("scene_%d" % args.scene)
            except Exception:      # noqa: BLE001
                e = None
            if e is not None:
                entry, entry_file = e, os.path.relpath(f, ROOT)
                break
        roof = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "G lane-op/s", "frac": None, "frac_arch": None, "traffic": None,
                "peak_arch": ARCH_PEAK_GLANEOPS,
                "peak_arch_source": "256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles; 157.3 TFLOP/s fp32 FMA / 2)"}
        roof["kernel_isa_sha256"] = my_hash
        roof["kernel"] = variant + ", %d waves per CU, %d of the inner records in LDS" % (plan["waves_per_cu"], plan["n_cached"])
        if entry is not None:
            lane_ops_per_ray = entry["lane_ops_per_ray"]
            pmc_hash = entry.get("kernel_isa_sha256")
            if my_hash is None:
                tie = "UNVERIFIED (%s)" % hash_note
            elif pmc_hash is None:
                tie = "UNVERIFIED (the imported PMC pass predates the hash tie)"
            elif pmc_hash == my_hash:
                tie = "current (PMC pass taken on this kernel binary: ISA sha256 %s)" % my_hash[:16]
            else:
                tie = "STALE (PMC pass was taken on kernel ISA %s, this library is %s)" % (pmc_hash[:16], my_hash[:16])
            roof["achieved"] = rays_per_launch_rank0 * lane_ops_per_ray / (kms * 1e-3) / 1e9
            roof["achieved_source"] = ("%s: useful VALU lane-ops per ray = %.1f (SQ_THREAD_CYCLES_VALU / rays, rocprofv3 --pmc pass of kernel %s, "
                                       "IMPORTED from %s) x %.4g rays per launch / %.2f ms render-kernel time, both measured in this run"
                                       % (tie, lane_ops_per_ray, entry.get("kernel", "?"), entry_file, rays_per_launch_rank0, kms))
            roof["lanes_per_valu_instruction"] = entry.get("lanes_per_valu_instruction")
            roof["wave_time_split"] = entry.get("wave_time_split")
            if entry.get("hbm_bytes_per_launch_1024spp") is not None and headline:
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
        roof["hbm"] = {"algorithmic_GBs": hbm_alg, "algorithmic_bytes_per_ray": b_ray, "measured_copy_peak_GBs": copy_gbs, "spec_peak_GBs": HBM_SPEC_GBS,
                       "frac_of_spec_peak": hbm_alg / HBM_SPEC_GBS,
                       "frac_of_measured_peak": (hbm_alg / copy_gbs) if copy_gbs else None,
                       "note": "SURVEY 8(d) figure V*64 + T*48 + 56 bytes per ray x rays per launch / kernel time.  A value above 1 means these bytes "
                               "never reach HBM: the inner tree is LDS resident and the rest is L2 resident (measured HBM traffic: roofline.traffic) -- "
                               "HBM is not the roof of this kernel, the vector issue port is"}
        t_d2h = time.perf_counter()
        fb = r.read_fb()                       # outside the timed region: image checksum, identical for every N
        d2h_ms = (time.perf_counter() - t_d2h) * 1e3
        checksum = int(sum(int(p.astype("int64").sum()) for p in fb))
        out = {
            "metric": "Mray/s", "value": mray, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (%d tris, %d BVH nodes), %dx%d, %d spp, depth %d" %
                                   (SCENE_NAMES.get(args.scene, "scene %d" % args.scene), scene.n_tris, scene.n_nodes, W, H, args.spp, args.depth),
                       "scene_id": args.scene,
                       "bvh": "SAH (exact sweep below 8192 triangles, 256 bins above), one triangle per leaf, children ordered by distance to the scene's default camera" if args.bvh == 1 else "reference builder (bvh/bvh.cu:206-346)",
                       "nan_direction_rays": "%.2f %% of the counted rays have a NaN direction (Sellmeier quirk Q1) and are answered 'miss' without walking the tree" % (100.0 * nan_c / rays_c),
                       "tiles": "8x8 px per wave, rank = tile % n_gpus", "gather": gather_via},
            "mpath_per_s": (W * H * args.spp * steps) / elapsed / 1e6,
            "rays_per_path": rays_per_path, "node_records_per_ray_V": V, "tri_tests_per_ray_T": T, "algorithmic_bytes_per_ray": b_ray,
            "kernel_ms_per_step": kms, "fb_checksum": checksum,
            # SURVEY 8(d)'s wall clock includes the hand-over of the framebuffer to the host; `value` / `ms_per_step` end with the
            # image in HBM (the boundary returns device memory), this adds the synchronous D2H of the three quantised planes
            "d2h_ms": d2h_ms, "ms_per_step_incl_d2h": elapsed / steps * 1e3 + d2h_ms,
            "mray_per_s_incl_d2h": total_rays / steps / (elapsed / steps + d2h_ms * 1e-3) / 1e6,
            "roofline": roof,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["gather_check"] = gather_check
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(srt, scene, cam, W, H, args.depth, args.bvh, gpu_renderer=r)
            except Exception as e:   # noqa: BLE001 -- the checker is optional for the measurement itself
                out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": usable_cores()[0], "kind": "port", "sample": "failed: %r" % (e,)}
            try:
                out["cpu_baseline"]["cfg1_single_thread"] = cfg1_single_thread(srt)
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"]["cfg1_single_thread"] = {"value": None, "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
