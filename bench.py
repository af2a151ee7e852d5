#!/usr/bin/env python3
"""bench.py -- headline benchmark of the spectral path-tracing hot path on MI355X.

Metric (BASELINE.json): Mray/s on the synthetic "random spheres" scene, 1920x1080, 1024 spp, depth 16
(config[2]; a ray = one closest-hit query).  One *step* = one full render of that image: every rank
renders its interleaved 8x8 tiles with the HIP kernel, ONE gather (RCCL over xGMI) brings the compact
tile buffers to rank 0, which scatters them into the block-linear planar framebuffer in HBM.  Scene,
camera and RNG states are resident in HBM before the timed region.  The image is fixed, so N GPUs split
the same work: scaling is "strong".

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the render kernel against HBM with the ALGORITHMIC bytes
per ray of SURVEY 8(d) (B_ray = V*64 + T*48 + 56; V, T measured by the instrumented kernel on the same
scene); `cpu_baseline` is the CPU oracle (a port: the reference has no CPU path) timed on this box's host
cores on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
NODE_BYTES, TRI_BYTES, MAT_BYTES = 64, 48, 56


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(srt, scene, cam, width, height, depth, mode, budget_s=20.0, gpu_renderer=None):
    """The oracle (kind "port") on all host cores, same scene/camera/size, reduced spp (rate is spp independent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as O
    cores = os.cpu_count() or 1
    osc = O.OracleScene(scene.triangles(), scene.materials(), scene.background())
    if mode == srt.BVH_REFERENCE:
        assert osc.build_reference(1984) == 1
    else:
        left, right, prim, _ = scene.bvh()
        assert osc.set_bvh(left, right, prim, 0) == 1
    t0 = time.time()
    r = osc.render(cam, width, height, 1, depth, threads=cores)           # calibration pass: 1 spp
    dt1 = max(time.time() - t0, 1e-3)
    spp = int(max(1, min(64, budget_s / dt1)))
    t0 = time.time()
    r = osc.render(cam, width, height, spp, depth, threads=cores)
    dt = time.time() - t0
    st = r["stats"]
    out = {"value": st["rays"] / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
           "sample": "%dx%d, %d spp, depth %d, same scene/camera/BVH, %d threads, %.1f s" % (width, height, spp, depth, cores, dt)}
    if gpu_renderer is not None:
        # the same frame at the same spp on the GPU: per-channel L-inf of the unquantised sRGB planes (BASELINE's parity
        # metric, target <= 1e-3) and the number of lanes whose bits differ -- the checker at work, outside the timed region
        import numpy as np
        # (instrumented kernel variant: keeps the production kernel's rocprof average to the timed launches)
        g = srt.render_image(scene, cam, width, height, spp, depth, renderer=gpu_renderer, count_traversal=True)
        linf = [float(np.max(np.abs(a - b))) for a, b in zip(g["lin"], r["lin"])]
        nbits = int(sum(int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))) for a, b in zip(g["xyz"], r["xyz"])))
        out["parity"] = {"linf_rgb": linf, "lanes_with_different_bits": nbits, "spp": spp, "rays_equal": bool(g["stats"]["rays"] == st["rays"])}
    return out


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
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N>1 path on ONE GPU: all ranks use cuda:0 and the gather goes through gloo/host (never used for reported numbers)")
    ap.add_argument("--pmc-traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: profiles/r01/hbm_traffic.json when the workload matches")
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
    if args.rehearse_gloo:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():      # launcher restricted each rank's visible devices to its own GPU
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

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

    # ---- V, T (node records / triangle tests per ray) from the instrumented kernel, outside the timed region
    r.init_device_params(W, H, 4, args.depth, 1984)
    r.set_count_traversal(True)
    r.render_chunk(W, H, 0, 0, stream)
    st = r.stats()
    red_dev = "cpu" if args.rehearse_gloo else "cuda"
    cnt = torch.tensor([st["rays"], st["node_visits"], st["tri_tests"], st["paths"]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(cnt)
    rays_c, V_c, T_c, paths_c = [float(x) for x in cnt.tolist()]
    V, T, rays_per_path = V_c / rays_c, T_c / rays_c, rays_c / paths_c
    b_ray = V * NODE_BYTES + T * TRI_BYTES + MAT_BYTES
    r.set_count_traversal(False)

    r.init_device_params(W, H, args.spp, args.depth, 1984)
    geom = dict(r.geom)

    local_tiles = None      # torch-owned staging tensor for the gather (multi-rank only)

    def step():
        # one step = one complete frame: seed the per-pixel RNG streams (init_device_params, rendering.cu:320-335), render,
        # assemble the framebuffer on rank 0.  Every step therefore produces the same image (fb_checksum).
        nonlocal local_tiles
        r.init_device_params(W, H, args.spp, args.depth, 1984)
        r.render_chunk(W, H, 0, 0, stream)
        if world == 1:
            r.scatter_tiles(None, stream)
        else:
            if local_tiles is None:
                _, _, _, tp = r.tile_buffer()
                local_tiles = torch.empty((tp, tiles.PLANES, tiles.LANES), dtype=torch.float32, device="cuda")
            r.copy_tile_buffer(local_tiles.data_ptr(), stream)        # stream-ordered D2D into the tensor RCCL sends
            if args.rehearse_gloo:
                gh = tiles.gather_tiles(local_tiles.cpu(), rank, world)
                g = gh.cuda() if rank == 0 else None
            else:
                g = tiles.gather_tiles(local_tiles, rank, world)      # the single collective of the path (RCCL over xGMI)
            if rank == 0:
                r.scatter_tiles(g.data_ptr(), stream)
                torch.cuda.current_stream().synchronize() if args.rehearse_gloo else None
        return r

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
    kernel_ms, rays_local = [], 0
    for k in range(args.steps):
        ts = time.time()
        step()
        # per-step kernel time from the HIP events the library records on `stream`; reading it waits for the
        # kernel only, and is inside the timed region on purpose (it costs one event sync).
        kernel_ms.append(r.last_kernel_ms())
        rays_local += r.stats()["rays"]
        if rank == 0:
            log("step %d/%d: kernel %.1f ms, wall %.2f s" % (k + 1, args.steps, kernel_ms[-1], time.time() - ts))
    barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(rays_local), elapsed, sum(kernel_ms) / max(len(kernel_ms), 1)], dtype=torch.float64, device=red_dev)
    if world > 1:
        rays_t = tot[0:1].clone(); dist.all_reduce(rays_t)
        mx = tot[1:3].clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        total_rays, elapsed, kms = float(rays_t[0]), float(mx[0]), float(mx[1])
    else:
        total_rays, elapsed, kms = float(tot[0]), float(tot[1]), float(tot[2])

    if rank == 0:
        steps = max(args.steps, 1)
        traffic = args.pmc_traffic_bytes
        if traffic is None and world == 1 and (args.scene, W, H, args.spp, args.depth, args.bvh) == (100, 1920, 1080, 1024, 16, 1):
            try:      # measured once with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on this exact workload
                traffic = json.load(open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic.json")))["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        mray = total_rays / elapsed / 1e6
        rays_per_launch_rank0 = rays_local / steps
        achieved = rays_per_launch_rank0 * b_ray / (kms * 1e-3) / 1e9            # GB/s, dominant kernel on this rank
        fb = r.read_fb()                       # outside the timed region: image checksum, identical for every N
        checksum = int(sum(int(p.astype("int64").sum()) for p in fb))
        out = {
            "metric": "Mray/s", "value": mray, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (%d tris, %d BVH nodes, %s tree), %dx%d, %d spp, depth %d" %
                                   ({0: "reference CORNELL scene", 1: "reference PRISM scene", 2: "reference TRIS scene", 100: "random-spheres tri scene",
                                     101: "100k-triangle mesh in the Cornell shell"}.get(args.scene, "scene %d" % args.scene), scene.n_tris, scene.n_nodes, "SAH" if args.bvh == 1 else "reference", W, H, args.spp, args.depth),
                       "scene_id": args.scene, "tiles": "8x8 px per wave, rank = tile % n_gpus", "gather": "1 RCCL gather of compact tiles"},
            "mpath_per_s": (W * H * args.spp * steps) / elapsed / 1e6,
            "rays_per_path": rays_per_path, "node_records_per_ray_V": V, "tri_tests_per_ray_T": T, "algorithmic_bytes_per_ray": b_ray,
            "kernel_ms_per_step": kms, "fb_checksum": checksum,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "note": "achieved = algorithmic bytes (V*64 + T*48 + 56 per ray) / render-kernel time. The scene is "
                                 "LDS/L2-resident, so these bytes never reach HBM (traffic = measured HBM bytes per launch, "
                                 "profiles/r01/hbm_traffic.json): the kernel is bound by VALU issue (96 % busy at 26 of 64 lanes per instruction, "
                                 "profiles/r01/v8_pmc_summary_64spp.txt), a frac above 1 is possible"},
        }
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(srt, scene, cam, W, H, args.depth, args.bvh, gpu_renderer=r)
            except Exception as e:   # the checker is optional for the measurement itself
                out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
