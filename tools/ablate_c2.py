"""Per-iteration cost of kernel variants on the Cornell box (round 6, C2's ceiling): for each library, the kernel time of an
800 x 800 x spp frame, the wave bounce-loop iterations it took and the lanes alive in them -> ns per wave-iteration.  A variant that
changes what paths do (an ablation that forces the light / cosine coin one way) changes the NUMBER of iterations; the cost of one is what
compares.  usage: python tools/ablate_c2.py [--spp 256] name=lib.so ..."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401
from raytracinginrust_amd import _lib, scenes
ap = argparse.ArgumentParser(); ap.add_argument('--spp', type=int, default=256); ap.add_argument('--rounds', type=int, default=4); ap.add_argument('libs', nargs='+')
a = ap.parse_args()
W = H = 800
rows = []
vs = []
for spec in a.libs:
    name, path = spec.split('=', 1)
    be = _lib.load_path(os.path.abspath(path)); b, cam, bg = scenes.cornell_box(be)
    vs.append((name, be, b, cam, bg))
res = {v[0]: [] for v in vs}
for r in range(a.rounds + 1):
    for name, be, b, cam, bg in vs:
        out = np.zeros((H, W, 3))
        assert be.lib.rt_render(b.h, C.byref(cam), (C.c_double * 3)(*bg), W, H, a.spp, 50, 0x5EED, 0, out.ctypes.data) == 0
        ms = C.c_float(); be.lib.rt_last_kernel_ms(b.h, C.byref(ms))
        st = (C.c_ulonglong * 3)(); be.lib.rt_last_stats(b.h, st)
        if r: res[name].append((ms.value, st[1], st[2], float(np.nanmean(out)) / a.spp))
for name, xs in res.items():
    ms = min(x[0] for x in xs); it, live, mean = xs[0][1], xs[0][2], xs[0][3]
    print(f'{name:12s} kernel {ms:8.3f} ms  wave-iterations {it:10d}  alive lanes {live / (64.0 * it):.4f}  ns per wave-iteration {ms * 1e6 / it:8.2f}  '
          f'iterations per sample {it * 64.0 / (W * H * a.spp):.3f}  frame mean {mean:.5f}')
