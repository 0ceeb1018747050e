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


class TileGatherer:
    """The tile path of one frame size, with its buffers allocated once: every rank renders its
    interleaved tiles into one packed buffer, rank 0 receives all buffers with ONE gather and
    scatters the tiles into the frame.

    device: where the packed buffers live -- "cuda" for RCCL (the render call writes device
    memory and the gather moves it over xGMI), "cpu" for gloo.  unpack: rtx_unpack_tiles or its
    numpy twin.  frame: optional caller-owned uint8 [height][width][3] array for rank 0's result.
    assemble (device buffers only): rtx_assemble_tiles as `assemble(device_index, pointers, width,
    height, tile_w, tile_h, frame)` -- the tiles are then scattered on the GPU and the frame comes
    down in one copy, instead of one copy and one host-side unpack per rank.
    """

    def __init__(self, width, height, tile_w, tile_h, rank, world, group=None, device="cpu",
                 unpack=unpack_tiles_numpy, frame=None, assemble=None):
        import torch
        self.geom = (width, height, tile_w, tile_h)
        self.rank, self.world, self.group, self.unpack = rank, world, group, unpack
        self.sizes = [packed_tile_bytes(width, height, tile_w, tile_h, r, world) for r in range(world)]
        pad = max(max(self.sizes), 1)
        self.mine = torch.zeros(pad, dtype=torch.uint8, device=device)
        self.parts = None
        self.stage = None
        self.assemble = assemble if self.mine.is_cuda else None
        if rank == 0:
            self.parts = [self.mine] if world == 1 else [torch.empty(pad, dtype=torch.uint8, device=device) for _ in range(world)]
            if self.mine.is_cuda and self.assemble is None:     # page-locked landing area for the device buffers
                self.stage = torch.empty((world, pad), dtype=torch.uint8).pin_memory()
            self.frame = frame if frame is not None else np.zeros((height, width, 3), dtype=np.uint8)

    def run(self, render_tiles):
        """render_tiles(first, stride, out) must fill the 1-D uint8 tensor `out` with the packed
        tiles first, first+stride, ...  Returns the frame on rank 0 and None elsewhere."""
        import torch
        import torch.distributed as dist
        width, height, tile_w, tile_h = self.geom
        if self.sizes[self.rank]:
            render_tiles(self.rank, self.world, self.mine[:self.sizes[self.rank]])
        if self.world > 1:
            dist.gather(self.mine, self.parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        if self.assemble is not None:
            torch.cuda.synchronize()                    # the gathered buffers are complete
            self.assemble(self.mine.device.index, [t.data_ptr() if n else 0 for t, n in zip(self.parts, self.sizes)],
                          width, height, tile_w, tile_h, self.frame)
            return self.frame
        if self.stage is not None:
            for r, t in enumerate(self.parts):
                if self.sizes[r]:
                    self.stage[r, :self.sizes[r]].copy_(t[:self.sizes[r]], non_blocking=True)
            torch.cuda.synchronize()
            host = [self.stage[r] for r in range(self.world)]
        else:
            host = self.parts
        for r, t in enumerate(host):
            if self.sizes[r]:
                self.unpack(t[:self.sizes[r]].numpy(), width, height, tile_w, tile_h, r, self.world, self.frame)
        return self.frame


def gather_tiled_frame(render_tiles, width, height, tile_w, tile_h, rank, world, group=None,
                       device="cpu", unpack=unpack_tiles_numpy):
    """Render this rank's tiles and assemble the frame on rank 0 (one-shot form of TileGatherer).

    render_tiles(first, stride, out) must fill the 1-D uint8 torch tensor `out` (on
    `device`) with the packed tiles first, first+stride, ...  Returns the uint8
    [height][width][3] frame on rank 0 and None elsewhere.
    """
    return TileGatherer(width, height, tile_w, tile_h, rank, world, group, device, unpack).run(render_tiles)
