"""Multi-GPU sharding of the render path: one process per GPU, torch.distributed.

Two ways the work splits (both replicate the scene on every GPU; secondary rays never
leave their GPU):

* whole frames -- animation frame f goes to rank f % world; no collective at all, every
  rank writes its own images (`shard_frames`);
* image tiles of one frame -- interleaved tiles, tile t goes to rank t % world for load
  balance (the reference's contiguous row bands, raytracer.cpp:107-126, are not);
  each rank renders its tiles into one packed buffer and rank 0 receives all buffers with
  ONE gather (RCCL when the group's backend is "nccl"), then scatters the tiles into the
  frame (`gather_tiled_frame`).  A 4K RGB8 frame is 24.9 MB, i.e. ~3.1 MB per rank at 8
  GPUs, arriving over 7 distinct xGMI links: a direct gather, never a ring.
"""
from __future__ import annotations

import numpy as np


def shard_frames(n_frames: int, rank: int, world: int):
    """Frame indices rendered by `rank` (round-robin keeps neighbouring, similarly
    expensive frames on different ranks)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def packed_tile_bytes(width, height, tile_w, tile_h, first, stride) -> int:
    tiles = ((width + tile_w - 1) // tile_w) * ((height + tile_h - 1) // tile_h)
    n_sel = 0 if first >= tiles else (tiles - first + stride - 1) // stride
    return n_sel * tile_w * tile_h * 3


def unpack_tiles_numpy(packed, width, height, tile_w, tile_h, first, stride, frame):
    """Pure-numpy twin of rtx_unpack_tiles (used where the HIP library is not loaded)."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    k = 0
    for tile in range(first, tiles_x * tiles_y, stride):
        x0, y0 = (tile % tiles_x) * tile_w, (tile // tiles_x) * tile_h
        blk = np.asarray(packed[k * tile_w * tile_h * 3:(k + 1) * tile_w * tile_h * 3]).reshape(tile_h, tile_w, 3)
        h, w = min(tile_h, height - y0), min(tile_w, width - x0)
        frame[y0:y0 + h, x0:x0 + w] = blk[:h, :w]
        k += 1
    return frame


def gather_tiled_frame(render_tiles, width, height, tile_w, tile_h, rank, world, group=None,
                       device="cpu", unpack=unpack_tiles_numpy):
    """Render this rank's tiles and assemble the frame on rank 0.

    render_tiles(first, stride, out) must fill the 1-D uint8 torch tensor `out` (on
    `device`) with the packed tiles first, first+stride, ...  Returns the uint8
    [height][width][3] frame on rank 0 and None elsewhere.
    """
    import torch
    import torch.distributed as dist
    sizes = [packed_tile_bytes(width, height, tile_w, tile_h, r, world) for r in range(world)]
    pad = max(max(sizes), 1)
    mine = torch.zeros(pad, dtype=torch.uint8, device=device)
    if sizes[rank]:
        render_tiles(rank, world, mine[:sizes[rank]])
    parts = [mine]
    if world > 1:
        bufs = [torch.empty(pad, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, bufs, dst=0, group=group)
        parts = bufs
    if rank != 0:
        return None
    frame = np.zeros((height, width, 3), dtype=np.uint8)
    for r, t in enumerate(parts):
        if sizes[r]:
            unpack(t[:sizes[r]].cpu().numpy(), width, height, tile_w, tile_h, r, world, frame)
    return frame
