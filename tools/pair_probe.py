"""Developer probe for the pair walk (RT_PAIR_BVH): on each scene the samples of the pair loop against the plain lock-step loop (differing
64-bit words), kernel times of default / lock-step / pair, and the walk's counters.   usage: python tools/pair_probe.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R, scenes
be = _lib.load()
earth = scenes.load_earthmap()
names = sys.argv[1:] or ['final', 'teapot', 'random', 'mesh0', 'mesh3']
def build(name):
    if name == 'final': return scenes.final_scene(be, *earth)
    if name == 'teapot': return scenes.cornell_test(be, scenes.asset_path('teapot.obj'))
    if name == 'random': return scenes.random_scene(be, aspect_ratio=1.0)
    from test_parity_gpu import _mesh_room
    return _mesh_room(be, int(name[4:]))
def stats(b):
    ts = R.last_traversal_stats(b)
    return (f'box steps {ts["traversal_steps"]}, lanes/step {ts["traversal_lanes"] / max(1, ts["traversal_steps"]):.1f}, arrivals resolved {ts["leaf_steps"]}, '
            f'pair walks {ts["leaf_lanes"]}')
for name in names:
    b, cam, bg = build(name)
    W, H, spp, depth = 96, 64, 8, 30
    _, ref = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_NO_SPECULATE_BVH, want_samples=True)
    _, got = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_PAIR_BVH, want_samples=True)
    bad = int((ref.view(np.uint64) != got.view(np.uint64)).sum())
    print(f'{name}: small frame, differing words {bad} of {ref.size}; {stats(b)}', flush=True)
    W = H = 400; spp = 64
    res = {}
    for tag, fl in (('default', 0), ('lockstep', R.RT_LOCKSTEP_BVH | R.RT_NO_SPECULATE_BVH), ('pair', R.RT_LOCKSTEP_BVH | R.RT_PAIR_BVH)):
        ms = []
        for _ in range(3):
            out = R.render(b, cam, bg, W, H, spp, 50, flags=fl)
            ms.append(R.last_kernel_ms(b))
        res[tag] = (min(ms), out)
        print(f'    {tag:9s} {min(ms):8.3f} ms  {W * H * spp / min(ms) / 1e3:8.1f} Msamples/s' + ('  ' + stats(b) if (tag == 'pair' or os.environ.get('PROBE_COUNT')) else ''), flush=True)
    d = np.abs(res['pair'][1] - res['lockstep'][1]); fin = np.isfinite(d)
    print(f'    frame sums pair vs lockstep: max |diff| {d[fin].max():.3e}, non-finite pattern equal: {bool(np.array_equal(np.isfinite(res["pair"][1]), np.isfinite(res["lockstep"][1])))}')
