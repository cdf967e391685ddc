"""The N>1 path on CPU: two gloo ranks shard the frame's tiles, one gather reassembles it (dist.py).
Tile contents come from the CPU oracle (this is a test), so the reassembled frame must equal the oracle's
full-frame render exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracinginrust_amd import dist as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, spp, depth, tile_px, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc
        from raytracinginrust_amd import scenes
        b, cam, bg = scenes.cornell_box(orc.load())
        n_local = D.n_local_tiles(W, H, tile_px, world)
        local = torch.zeros((n_local, tile_px, 3), dtype=torch.float64)
        rows_per_tile = tile_px // W
        for slot, t in enumerate(D.local_tile_ids(W, H, tile_px, rank, world)):
            r0 = t * rows_per_tile
            if r0 >= H:
                continue                                   # padding tile
            r1 = min(H, r0 + rows_per_tile)
            img = orc.render(b, cam, bg, W, H, spp, depth, rows=(r0, r1), nthreads=1)
            flat = torch.from_numpy(img[r0:r1].reshape(-1, 3))
            local[slot, : flat.shape[0]] = flat
        frame = D.gather_frame(local, W, H, tile_px, dst=0)
        if rank == 0:
            full = orc.render(b, cam, bg, W, H, spp, depth, nthreads=1)
            q.put(bool(np.array_equal(frame.numpy(), full)))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 10), (2, 9), (3, 8)])
def test_tile_shard_and_gather_gloo(world, H):
    W, spp, depth = 12, 2, 8
    tile_px = 2 * W                                       # two rows per tile; H = 9 leaves a ragged last tile, world 3 a padded one
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, spp, depth, tile_px, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_assemble_unpermutes():
    W, H, tile_px, world = 5, 3, 4, 2           # 15 pixels -> 4 tiles -> 2 per rank
    n_local = D.n_local_tiles(W, H, tile_px, world)
    g = torch.zeros((world, n_local, tile_px, 3), dtype=torch.float64)
    for r in range(world):
        for slot, t in enumerate(D.local_tile_ids(W, H, tile_px, r, world)):
            for k in range(tile_px):
                g[r, slot, k, :] = t * tile_px + k
    frame = D.assemble(g, W, H, tile_px)
    assert torch.equal(frame[..., 0].reshape(-1), torch.arange(W * H, dtype=torch.float64))
