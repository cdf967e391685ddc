"""One process per GPU: image tiles shard across ranks, one gather (RCCL over xGMI) at frame end.

The frame's W*H pixels (output order) are cut into tiles of `tile_px` consecutive pixels; rank r of N
renders tiles t with t % N == r (interleaved, so expensive image regions spread over all GPUs) into a
packed device buffer, and a single `torch.distributed.gather` moves the N buffers to rank 0, which
un-permutes them into the frame.  There is no other data-path collective: pixels are independent and
the scene (< 2 MB) is replicated.  `torch.distributed` with backend "nccl" is RCCL on ROCm; the same
code runs over "gloo" with CPU tensors for the multi-process tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import render as R

# Tiles are dealt round-robin (tile % world == rank).  With a power-of-two tile size, frames whose row holds a multiple of
# `world` tiles (3840 / 64 = 60, 60 % 8 = 4 ...) put the same image columns on the same ranks every (other) row: measured
# 8-way balance 0.978 at 3840x2160 with 64-pixel tiles.  A prime tile size lets tile boundaries drift across rows: 0.998.
DEFAULT_TILE_PX = 67


def n_tiles(W: int, H: int, tile_px: int) -> int:
    return (W * H + tile_px - 1) // tile_px


def n_local_tiles(W: int, H: int, tile_px: int, world: int) -> int:
    """Tiles per rank, padded so that it is the same on every rank (a plain gather needs equal counts)."""
    return (n_tiles(W, H, tile_px) + world - 1) // world


def local_tile_ids(W: int, H: int, tile_px: int, rank: int, world: int):
    """Global tile index of each local slot q (may point past the last real tile: padding)."""
    return [rank + q * world for q in range(n_local_tiles(W, H, tile_px, world))]


def assemble(gathered: torch.Tensor, W: int, H: int, tile_px: int) -> torch.Tensor:
    """[world, n_local, tile_px, 3] rank-major tile buffers -> (H, W, 3) frame."""
    world, n_local = gathered.shape[0], gathered.shape[1]
    frame = gathered.permute(1, 0, 2, 3).reshape(n_local * world * tile_px, 3)
    return frame[: W * H].reshape(H, W, 3)


def gather_frame(local: torch.Tensor, W: int, H: int, tile_px: int, dst: int = 0, gathered: torch.Tensor | None = None):
    """`local`: this rank's [n_local, tile_px, 3] buffer.  Returns the (H, W, 3) frame on `dst`, None elsewhere.  The one collective
    writes straight into the rows of `gathered` ([world, n_local, tile_px, 3], reusable; allocated here if not given), and the frame
    is ONE strided copy out of it (the un-permute): no intermediate stack of the per-rank buffers."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return assemble(local.unsqueeze(0), W, H, tile_px)
    staged = local
    if local.is_cuda and dist.get_backend() == "gloo":      # gloo (CPU tests / single-GPU development) gathers host tensors
        staged = local.cpu()
    bufs = None
    if rank == dst:
        if gathered is None or gathered.device != staged.device or tuple(gathered.shape) != (world,) + tuple(staged.shape):
            gathered = torch.empty((world,) + tuple(staged.shape), dtype=staged.dtype, device=staged.device)
        bufs = list(gathered.unbind(0))                      # views: the gather's receive buffers ARE the rows
    dist.gather(staged, bufs, dst=dst)
    if rank != dst:
        return None
    return assemble(gathered, W, H, tile_px).to(local.device)


class TileRenderer:
    """Per-rank render state: the replicated scene and reusable device buffers for this rank's tiles.

    `pipeline` = frames kept in flight (1 or 2).  With 2, consecutive frames alternate between two HIP streams and two tile
    buffers, so that the start of frame i+1 overlaps the drain of frame i (a persistent launch spends ~0.5 ms filling up and
    emptying out, *measured*: 16.17 -> 15.73 ms per 1/8 share of the C2 frame, 4.41 -> 4.08 ms per 1/32; no gain at whole-frame
    size); the gather of frame i is ordered after its own kernel on its own stream.  Not the bench default: with a real gather the
    collective's kernels have to find CU room beside a persistent kernel that fills the chip, which was not measurable on one GPU.  A returned frame is valid once that stream is done: call `sync()` (or synchronise the device)
    before reading it, and copy it if it must outlive the next `pipeline` frames."""

    def __init__(self, builder, cam, background, W, H, spp, max_depth, seed=0x5EED, flags=R.RT_F64, tile_px=DEFAULT_TILE_PX,
                 rank=None, world=None, device=None, pipeline=1):
        self.b, self.cam, self.bg = builder, cam, background
        self.W, self.H, self.spp, self.max_depth, self.seed, self.flags, self.tile_px = W, H, spp, max_depth, seed, flags, tile_px
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.n_local = n_local_tiles(W, H, tile_px, self.world)
        assert self.n_local == R.local_tiles(W, H, tile_px, self.rank, self.world)
        assert pipeline in (1, 2)
        self.pipeline = pipeline
        self.locals = [torch.empty((self.n_local, tile_px, 3), dtype=torch.float64, device=self.device) for _ in range(pipeline)]
        self.local = self.locals[0]
        # different priorities = different hardware queues: with two equal-priority streams the runtime sometimes put both on one
        # queue and the launches did not overlap at all (measured)
        self.streams = [torch.cuda.Stream(self.device, priority=-(k % 2)) for k in range(pipeline)] if pipeline > 1 else []
        self.frames = 0
        self._gathered = []            # rank `dst` only: receive buffers of the gather, one per frame in flight

    def render_local(self, buf=None) -> torch.Tensor:
        """Launch the path-tracing kernel for this rank's tiles on torch's current stream (asynchronous)."""
        buf = self.local if buf is None else buf
        stream = torch.cuda.current_stream(self.device).cuda_stream
        R.render_tiles_device(self.b, self.cam, self.bg, self.W, self.H, self.spp, self.max_depth, self.seed, self.flags,
                              self.tile_px, self.rank, self.world, buf.data_ptr(), buf.numel() * buf.element_size(), stream)
        return buf

    def _frame(self, buf, dst):
        local = self.render_local(buf)
        if self.world == 1:
            return assemble(local.unsqueeze(0), self.W, self.H, self.tile_px)
        if self.rank == dst and not (local.is_cuda and dist.get_backend() == "gloo"):
            k = self.frames % self.pipeline if self.pipeline > 1 else 0      # one receive buffer per frame in flight
            if len(self._gathered) <= k:
                self._gathered.extend([None] * (k + 1 - len(self._gathered)))
            if self._gathered[k] is None:
                self._gathered[k] = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
            return gather_frame(local, self.W, self.H, self.tile_px, dst, self._gathered[k])
        return gather_frame(local, self.W, self.H, self.tile_px, dst)

    def render_frame(self, dst: int = 0):
        """One frame: local tiles, then the single gather.  Returns the (H, W, 3) per-pixel sums on `dst`."""
        k = self.frames % self.pipeline
        self.frames += 1
        if self.pipeline == 1:
            return self._frame(self.locals[0], dst)
        side = self.streams[k]
        side.wait_stream(torch.cuda.current_stream(self.device))       # whatever the caller queued so far comes first
        with torch.cuda.stream(side):
            return self._frame(self.locals[k], dst)

    def sync(self):
        """Wait for every frame in flight."""
        for s in self.streams:
            s.synchronize()
        torch.cuda.current_stream(self.device).synchronize()
