import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracinginrust_amd import _lib, dist as D, render as R, scenes
be = _lib.load()
b, cam, bg = scenes.cornell_box(be, aspect_ratio=3840 / 2160)
W, H, spp, depth = 3840, 2160, 4, 50
full = R.render(b, cam, bg, W, H, spp, depth)
print('full 4K mean', full.mean() / spp, 'nonfinite', R.last_stats(b))
parts = []
for rank in range(8):
    tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, tile_px=64, rank=rank, world=8)
    p = tr.render_local().clone(); torch.cuda.synchronize(); parts.append(p)
    print('rank', rank, 'mean', float(p.mean()) / spp)
frame = D.assemble(torch.stack(parts, 0), W, H, 64).cpu().numpy()
print('assembled mean', frame.mean() / spp, 'max abs diff vs full', np.abs(frame - full).max())
small = R.render(b, cam, bg, W // 8, H // 8, 64, depth)
print('480x270x64 mean', small.mean() / 64)
c2 = scenes.cornell_box(be); sq = R.render(c2[0], c2[1], c2[2], 400, 400, 64, depth); print('square 400x400 mean', sq.mean() / 64)
