"""world_size-2 rehearsal of the multi-GPU sharding on CPU (gloo): the tile split + gather
must reproduce the single-process frame byte for byte, and frame sharding must cover every
frame exactly once.  The renderer behind the sharding logic is the CPU oracle here (this is
a test; on GPUs it is the HIP core)."""
import os
import socket

import numpy as np
import pytest

import _checkers as ck
import raytracer_amd as rt
from raytracer_amd import distributed as D
from raytracer_amd import scenes


def test_frame_sharding_partitions_frames():
    for world in (1, 2, 3, 8):
        seen = sorted(f for r in range(world) for f in D.shard_frames(360, r, world))
        assert seen == list(range(360))
    with pytest.raises(ValueError):
        D.shard_frames(10, 2, 2)


def test_tile_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    w, h, tw, th = 37, 23, 8, 5
    frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    for world in (1, 2, 3, 5):
        out = np.zeros_like(frame)
        for r in range(world):
            packed = _pack(frame, tw, th, r, world)
            assert packed.size == D.packed_tile_bytes(w, h, tw, th, r, world)
            D.unpack_tiles_numpy(packed, w, h, tw, th, r, world, out)
        assert np.array_equal(out, frame)


def test_c_unpack_matches_numpy():
    rng = np.random.default_rng(1)
    w, h, tw, th = 50, 31, 16, 8
    frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    a, b = np.zeros_like(frame), np.zeros_like(frame)
    for r in range(3):
        packed = _pack(frame, tw, th, r, 3)
        D.unpack_tiles_numpy(packed, w, h, tw, th, r, 3, a)
        rt.unpack_tiles(packed, w, h, tw, th, r, 3, b)      # host helper of the device library
    assert np.array_equal(a, frame) and np.array_equal(b, frame)


def _pack(frame, tw, th, first, stride):
    h, w, _ = frame.shape
    tiles_x, tiles_y = (w + tw - 1) // tw, (h + th - 1) // th
    out = []
    for tile in range(first, tiles_x * tiles_y, stride):
        x0, y0 = (tile % tiles_x) * tw, (tile // tiles_x) * th
        blk = np.zeros((th, tw, 3), dtype=np.uint8)
        sub = frame[y0:y0 + th, x0:x0 + tw]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        out.append(blk.reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


def _worker(rank, world, port, scene_path, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = rt.HostScene(scene_path)
    cam, _ = hs.camera(0)
    full, _, _, _ = ck.oracle_render(hs.desc, cam, want_hits=False, nthreads=2)

    def render_tiles(first, stride, out):
        out.copy_(torch.from_numpy(_pack(full, 16, 8, first, stride)))

    frame = D.gather_tiled_frame(render_tiles, cam.width, cam.height, 16, 8, rank, world)
    if rank == 0:
        np.save(out_path, frame)
    # whole-frame sharding: every rank reports which frames it owns
    mine = torch.zeros(12, dtype=torch.int64)
    for f in D.shard_frames(12, rank, world):
        mine[f] = 1
    dist.all_reduce(mine)
    assert mine.tolist() == [1] * 12
    dist.destroy_process_group()


def test_tile_gather_world2_gloo(unit_scenes, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, unit_scenes["prims"], out), nprocs=2, join=True)
    hs = rt.HostScene(unit_scenes["prims"])
    cam, _ = hs.camera(0)
    full, _, _, _ = ck.oracle_render(hs.desc, cam, want_hits=False)
    assert np.array_equal(np.load(out), full)
