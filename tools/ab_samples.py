"""Per-sample comparison of kernel builds in ONE process: every sample of a small frame must be bit-identical between the builds (a
scheduling or code-generation change must not move a single bit).  usage: python tools/ab_samples.py [--scene cornell] name=lib.so name=lib.so ..."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401
from raytracinginrust_amd import _lib, scenes
ap = argparse.ArgumentParser(); ap.add_argument('--scene', default='cornell'); ap.add_argument('--size', type=int, default=128); ap.add_argument('--spp', type=int, default=32)
ap.add_argument('libs', nargs='+'); a = ap.parse_args()
def build(be):
    if a.scene == 'cornell': return scenes.cornell_box(be)
    if a.scene == 'smoke': return scenes.cornell_box_with_smoke(be)
    if a.scene == 'random': return scenes.random_scene(be, aspect_ratio=1.0)
    if a.scene == 'teapot': return scenes.cornell_test(be, scenes.asset_path('teapot.obj'))
    return scenes.final_scene(be, *scenes.load_earthmap())
ref = None
for spec in a.libs:
    name, path = spec.split('=', 1)
    be = _lib.load_path(os.path.abspath(path))
    b, cam, bg = build(be)
    W = H = a.size
    out = np.zeros((H, W, 3)); smp = np.zeros((H, W, a.spp, 3))
    rc = be.lib.rt_render_samples(b.h, C.byref(cam), (C.c_double * 3)(*bg), W, H, a.spp, 50, 0x5EED, 0, out.ctypes.data, smp.ctypes.data)
    assert rc == 0, be.lib.rt_last_error()
    if ref is None: ref = smp; print(f'{name}: reference, {smp.size} values, {int((~np.isfinite(smp)).sum())} non-finite')
    else: print(f'{name}: differing 64-bit words vs the first build: {int((ref.view(np.uint64) != smp.view(np.uint64)).sum())}')
