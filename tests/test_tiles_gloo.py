"""The N > 1 path on CPU: tile partition + the single gather + scatter, world_size 2 and 3 over gloo.
The per-rank renderer stand-in is the oracle (allowed in tests); everything else is the product's host logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, depth, groups, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
    tiles = srt.tiles
    import oracle_binding as O
    from helpers import oracle_scene_for
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = srt.Scene.builtin(srt.SCENE_PRISM).build_bvh(srt.BVH_REFERENCE, 1984)
    cam = scene.default_camera(W, H)
    osc = oracle_scene_for(O, scene, 0)
    tx, ty = 28, 16
    bx, by = W // tx + 1, H // ty + 1
    full = osc.render(cam, W, H, spp, depth, threads=2)              # stand-in renderer: every pixel, then keep own tiles
    planes = np.stack(list(full["fb"]) + list(full["lin"]) + list(full["xyz"]))      # [9, n_lanes] block-linear
    g = tiles.tile_geometry(W, H, tx, ty, bx, by, world)
    # a rank's tile buffer: [group][tile][plane of the group][lane]; `groups` of them travel (1 = the quantised framebuffer)
    local = torch.zeros((tiles.GROUPS, g["tiles_padded"], tiles.GROUP_PLANES, tiles.LANES), dtype=torch.float32)
    lane = torch.arange(tiles.LANES)
    for k, t in enumerate(tiles.local_tile_ids(g["n_tiles"], rank, world)):
        i = (t % g["tiles_x"]) * 8 + lane % 8
        j = (t // g["tiles_x"]) * 8 + lane // 8
        ok = (i < g["cover_w"]) & (j < g["cover_h"])
        idx = tiles.block_linear_index(i, j, tx, ty, bx)
        vals = torch.from_numpy(planes)[:, idx[ok]]                  # [9, n_ok]
        for grp in range(tiles.GROUPS):
            local[grp, k][:, ok] = vals[3 * grp: 3 * grp + 3]
    gathered = tiles.gather_tiles(local[:groups].contiguous(), rank, world)                 # ONE collective
    if rank == 0:
        fb = tiles.scatter_tiles_torch(gathered, W, H, tx, ty, bx, by, world)
        np.save(out_path, np.concatenate([fb.numpy(), planes]))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,groups", [(2, 3), (3, 3), (2, 1)])
def test_partition_gather_scatter_gloo(world, groups, tmp_path):
    """groups = 3: all nine planes travel (parity tests); groups = 1: the default exchange unit, the quantised framebuffer."""
    W, H, spp, depth = 50, 37, 2, 4
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, depth, groups, out), nprocs=world, join=True)
    both = np.load(out)
    fb, planes = both[:3 * groups], both[3 * groups:]
    # every in-image lane of every gathered plane arrives exactly once and unchanged; lanes outside the image stay 0
    assert planes.shape[0] == 9
    assert np.array_equal(fb.view(np.uint32), planes[:3 * groups].view(np.uint32))


def test_tile_geometry_and_ownership(srt):
    tiles = srt.tiles
    for (W, H, world) in [(1920, 1080, 8), (1280, 720, 3), (50, 37, 2), (8, 8, 4)]:
        bx, by = W // 28 + 1, H // 16 + 1
        g = tiles.tile_geometry(W, H, 28, 16, bx, by, world)
        assert g["n_tiles"] == ((28 * bx + 7) // 8) * ((16 * by + 7) // 8)     # the whole reference grid, not the chunk
        owned = sorted(t for r in range(world) for t in tiles.local_tile_ids(g["n_tiles"], r, world))
        assert owned == list(range(g["n_tiles"]))                      # a partition: no tile twice, none missing
        sizes = [len(tiles.local_tile_ids(g["n_tiles"], r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) <= g["tiles_padded"]
    # block-linear index is a bijection pixel -> lane (rendering.cu:156-165)
    W, H = 61, 35
    bx = W // 28 + 1
    i, j = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    idx = tiles.block_linear_index(i, j, 28, 16, bx)
    assert idx.unique().numel() == W * H
