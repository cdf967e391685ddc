"""Kernel time of very small renders: the fixed start/drain cost of one persistent launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, render as R, scenes
be = _lib.load(); b, cam, bg = scenes.cornell_box(be)
for (W, H, spp, depth) in ((8, 8, 1, 50), (64, 64, 1, 50), (64, 64, 64, 50), (800, 800, 1, 50), (800, 800, 1, 1), (800, 800, 4, 50), (800, 800, 16, 50)):
    ts = []
    for _ in range(5):
        R.render(b, cam, bg, W, H, spp, depth); ts.append(R.last_kernel_ms(b))
    print(f'{W}x{H}x{spp} depth {depth}: {min(ts[1:]) * 1e3:8.1f} us   ({W * H * spp / 5246e6 * 1e6:8.1f} us at the full-frame rate)')
