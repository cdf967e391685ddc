"""Per-rank kernel time of a frame split over `world` ranks (all ranks rendered in turn on one GPU): load balance of the
tile interleave.  usage: python tools/rank_balance.py W H spp world tile_px [tile_px ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, dist as D, render as R, scenes
W, H, spp, world = [int(x) for x in sys.argv[1:5]]
be = _lib.load(); b, cam, bg = scenes.cornell_box(be, aspect_ratio=W / H)
for tile_px in [int(x) for x in sys.argv[5:]]:
    ms = []
    for rank in range(world):
        tr = D.TileRenderer(b, cam, bg, W, H, spp, 50, tile_px=tile_px, rank=rank, world=world)
        ts = []
        for _ in range(3):
            tr.render_local(); torch.cuda.synchronize(); ts.append(R.last_kernel_ms(b))
        ms.append(min(ts[1:]))
    print(f'{W}x{H}x{spp} world {world} tile_px {tile_px:4d}: per-rank ms min {min(ms):8.3f} max {max(ms):8.3f} mean {sum(ms)/len(ms):8.3f}  balance (mean/max) {sum(ms)/len(ms)/max(ms):.4f}')
