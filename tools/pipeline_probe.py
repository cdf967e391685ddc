"""Frames in flight: time per frame share with TileRenderer(pipeline=1) vs (pipeline=2) — does the start of frame i+1 hide
the drain of frame i?  usage: python tools/pipeline_probe.py [world] (rank 0's share of the C2 frame, one GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, dist as D, render as R, scenes
be = _lib.load()
W = H = 800; spp = 1024; depth = 50; world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b, cam, bg = scenes.cornell_box(be)
D.gather_frame = lambda local, *a, **k: local      # no process group here: time the render side only
def run(tr, n):
    for _ in range(3): tr.render_frame()
    tr.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.render_frame()
    tr.sync(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
n = max(8, 5 * world)
for rep in range(2):
    for p in (1, 2):
        tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, rank=0, world=world, pipeline=p)
        R.kernel_time_total(b, reset=True)
        ms = run(tr, n)
        tot, cnt = R.kernel_time_total(b)
        print(f'world {world}: {p} frame(s) in flight: {ms:8.3f} ms per frame share   (mean kernel duration by events {tot / cnt:8.3f} ms)')
