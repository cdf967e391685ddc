"""Developer probe: the persistent-traversal schedule (start a traversal pass at `hi` lanes inside a BVH, stop below `lo`, leaf step once
leaf/64 of the walking lanes hold a leaf) on the teapot room.   usage: python tools/trav_sweep.py"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
w = workloads.WORKLOADS['C4']
b, cam, bg = workloads.build(w, be, None)
def t(hi, lo, leaf):
    R.set_traversal_schedule(b, hi, lo, leaf)
    ms = []
    for _ in range(3):
        R.render(b, cam, bg, 960, 540, 64, 50); ms.append(R.last_kernel_ms(b))
    return min(ms)
base = t(56, 16, 32); print(f'(56, 16, 32): {base:.3f} ms', flush=True)
res = []
for hi, lo, leaf in itertools.product((32, 40, 48, 56), (12, 16, 24, 32), (16, 24, 32)):
    if lo > hi: continue
    ms = t(hi, lo, leaf); res.append((ms, hi, lo, leaf)); print(f'({hi}, {lo}, {leaf}): {ms:.3f} ms  {base / ms:.3f}x', flush=True)
res.sort(); print('best:', res[:5])
