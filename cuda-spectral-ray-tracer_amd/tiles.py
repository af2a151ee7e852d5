"""Tile partition + framebuffer gather for the multi-GPU path (one process per GPU, torch.distributed).

The image chunk is cut into 8x8-pixel tiles (one wave each).  Rank r of W renders tiles t with t % W == r:
interleaving at tile granularity balances the sky / geometry cost gradient, and because every pixel's RNG
seed is 1984 + its block-linear index (rendering/rendering.cu:137) the image is bit-identical for any W.
The only exchange step is ONE gather of the compact tile buffers to rank 0 (RCCL over xGMI when the
tensors are on GPUs; gloo in the CPU tests), followed by a scatter into the block-linear planar framebuffer.
"""
import torch
import torch.distributed as dist

TILE = 8
PLANES = 9          # quantised r,g,b | unquantised sRGB r,g,b | XYZ sums
GROUP_PLANES = 3    # ... in three groups of three; a rank's tile buffer is [group][tile][plane of the group][lane]
GROUPS = 3          # the gather moves the first group (the quantised framebuffer, 12 B / pixel) or all three (parity tests)
LANES = 64


def tile_geometry(width, height, tx, ty, bx, by, world):
    """Tiles cover the whole reference grid (tx*bx x ty*by pixels), NOT just the current chunk: the tile number of a pixel --
    and so the rank that owns its persistent RNG stream -- is the same for every chunk of a render (srt_render_chunk).
    cover_w / cover_h are the pixels of the chunk that exist."""
    cover_w, cover_h = min(width, tx * bx), min(height, ty * by)
    tiles_x, tiles_y = (tx * bx + TILE - 1) // TILE, (ty * by + TILE - 1) // TILE
    n_tiles = tiles_x * tiles_y
    return dict(tiles_x=tiles_x, tiles_y=tiles_y, n_tiles=n_tiles, tiles_padded=(n_tiles + world - 1) // world,
                cover_w=cover_w, cover_h=cover_h)


def local_tile_ids(n_tiles, rank, world):
    return list(range(rank, n_tiles, world))


def gather_tiles(local, rank, world, group=None, dst=0):
    """local: [groups, tiles_padded, GROUP_PLANES, LANES] float32 tensor (same shape on every rank; groups = 1 or 3).
    Returns [world, groups, tiles_padded, GROUP_PLANES, LANES] on rank `dst`, None elsewhere.  One collective."""
    if world == 1:
        return local.unsqueeze(0)
    if rank == dst:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(local, None, dst=dst, group=group)
    return None


def block_linear_index(i, j, tx, ty, bx):
    """idx = ty*28+tx + 448*(by*gridDim.x+bx), rendering/rendering.cu:156-165 (i, j chunk-relative tensors)."""
    gbx, gby = i // tx, j // ty
    return (j - gby * ty) * tx + (i - gbx * tx) + tx * ty * (gby * bx + gbx)


def scatter_tiles_torch(gathered, width, height, tx, ty, bx, by, world):
    """Pure-torch scatter of gathered tiles into block-linear planes (used by the CPU/gloo tests; the GPU path uses
    srt_scatter_tiles).  gathered: [world, groups, tiles_padded, GROUP_PLANES, LANES]; returns [groups * 3, n_lanes]."""
    g = tile_geometry(width, height, tx, ty, bx, by, world)
    n_lanes = tx * ty * bx * by
    groups = gathered.shape[1]
    fb = torch.zeros((groups * GROUP_PLANES, n_lanes), dtype=gathered.dtype, device=gathered.device)
    t = torch.arange(g["n_tiles"], device=gathered.device)
    lane = torch.arange(LANES, device=gathered.device)
    i = (t % g["tiles_x"])[:, None] * TILE + (lane % TILE)[None, :]
    j = (t // g["tiles_x"])[:, None] * TILE + (lane // TILE)[None, :]
    ok = (i < g["cover_w"]) & (j < g["cover_h"])
    idx = block_linear_index(i, j, tx, ty, bx)
    for grp in range(groups):
        src = gathered[t % world, grp, t // world]            # [n_tiles, GROUP_PLANES, LANES]
        for p in range(GROUP_PLANES):
            fb[grp * GROUP_PLANES + p][idx[ok]] = src[:, p, :][ok]
    return fb
