"""A/B timing of kernel variants in ONE process, interleaved rounds (cdna guide §5.4 rule 24).
usage: python tools/ab.py [--spp 256] [--rounds 5] [--scene cornell] name=path/to/lib.so[@flags[@ENV=V;ENV2=V]] ...
(per-variant render flags are OR-ed onto --flags; per-variant environment knobs are set around that variant's renders)
Prints per variant: min / median kernel ms and Msamples/s, and checks variants agree with the first one."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime in the process)
from raytracinginrust_amd import _lib, scenes
from raytracinginrust_amd.api import CameraParams

ap = argparse.ArgumentParser()
ap.add_argument('--spp', type=int, default=256); ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--scene', default='cornell'); ap.add_argument('--size', type=int, default=800); ap.add_argument('--depth', type=int, default=50)
ap.add_argument('--flags', type=int, default=0)
ap.add_argument('libs', nargs='+')
a = ap.parse_args()

def build(be):
    if a.scene == 'cornell': return scenes.cornell_box(be)
    if a.scene == 'random': return scenes.random_scene(be, aspect_ratio=1.0)
    if a.scene == 'teapot': return scenes.cornell_test(be, scenes.asset_path('teapot.obj'))
    if a.scene == 'final':
        return scenes.final_scene(be, *scenes.load_earthmap())

variants = []
_loaded = {}
for spec in a.libs:
    name, path = spec.split('=', 1) if '=' in spec else (os.path.basename(spec), spec)
    parts = path.split('@')
    path, vflags = parts[0], int(parts[1]) if len(parts) > 1 and parts[1] else 0
    venv = dict(kv.split('=') for kv in parts[2].split(';')) if len(parts) > 2 and parts[2] else {}
    path = os.path.abspath(path)
    be = _loaded.get(path) or _lib.load_path(path)
    _loaded[path] = be
    b, cam, bg = build(be)
    variants.append((name, be, b, cam, bg, vflags, venv))
W = H = a.size
def run(v):
    name, be, b, cam, bg, vflags, venv = v
    out = np.zeros((H, W, 3))
    for k, val in venv.items(): os.environ[k] = val
    rc = be.lib.rt_render(b.h, C.byref(cam), (C.c_double * 3)(*bg), W, H, a.spp, a.depth, 0x5EED, a.flags | vflags, out.ctypes.data)
    for k in venv: os.environ.pop(k, None)
    assert rc == 0, be.lib.rt_last_error()
    ms = C.c_float(); be.lib.rt_last_kernel_ms(b.h, C.byref(ms))
    return out, ms.value
ref = None; times = {v[0]: [] for v in variants}
for r in range(a.rounds + 1):
    for v in variants:
        out, ms = run(v)
        if r == 0:
            if ref is None: ref = out
            else:
                d = np.abs(out - ref); print(f'{v[0]}: max |diff| vs {variants[0][0]} = {np.nanmax(d):.3e} (rel {np.nanmax(d / (np.abs(ref) + 1e-300)):.2e})')
        else: times[v[0]].append(ms)
for name, ts in times.items():
    ts = sorted(ts); print(f'{name:24s} min {ts[0]:8.3f} ms  median {ts[len(ts)//2]:8.3f} ms  {W*H*a.spp/ts[0]/1e3:9.1f} Msamples/s')
