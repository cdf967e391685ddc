"""Kernel time of one rank's share of the C2 frame (tiles t % world == rank) vs the whole frame: shows the end-of-frame
tail a strong-scaled run pays.  RT_AMD_LIB selects the library build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, dist as D, render as R, scenes
be = _lib.load(); b, cam, bg = scenes.cornell_box(be)
W = H = 800; spp = 1024; depth = 50
res = {}
for world in (1, 2, 4, 8, 16, 32, 64, 256):
    tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, tile_px=64, rank=0, world=world)
    ts = []
    for _ in range(4):
        tr.render_local(); torch.cuda.synchronize(); ts.append(R.last_kernel_ms(b))
    res[world] = min(ts[1:])
for world, ms in res.items():
    print(f'world {world}: rank-0 kernel {ms:8.3f} ms   ideal {res[1]/world:8.3f}   efficiency {res[1]/world/ms:.4f}')
