"""Developer probe for the lane-cooperative BVH walk (RT_JOINT_BVH): on each scene the samples of the joint loop against the plain
lock-step loop (differing 64-bit words), the kernel times of both (and of the default loop), and the walk's counters (box steps, lanes
per step, hand-overs, redone rays).   usage: python tools/joint_probe.py [scene ...]   (RT_AMD_LIB selects the build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R, scenes
be = _lib.load()
earth = scenes.load_earthmap()
names = sys.argv[1:] or ['final', 'mesh0', 'mesh3', 'mesh7']
def build(name):
    if name == 'final': return scenes.final_scene(be, *earth)
    if name == 'teapot': return scenes.cornell_test(be, scenes.asset_path('teapot.obj'))
    if name == 'random': return scenes.random_scene(be, aspect_ratio=1.0)
    from test_parity_gpu import _mesh_room
    return _mesh_room(be, int(name[4:]))
for name in names:
    b, cam, bg = build(name)
    W, H, spp, depth = 96, 64, 8, 30
    _, ref = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_NO_SPECULATE_BVH, want_samples=True)
    _, got = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_JOINT_BVH, want_samples=True)
    ts = R.last_traversal_stats(b)
    bad = int((ref.view(np.uint64) != got.view(np.uint64)).sum())
    print(f'{name}: small frame, differing words {bad} of {ref.size}; box steps {ts["traversal_steps"]}, lanes/step {ts["traversal_lanes"] / max(1, ts["traversal_steps"]):.1f}, '
          f'hand-overs {ts["leaf_steps"]}, redone rays {ts["leaf_lanes"]}', flush=True)
    W = H = 400; spp = 64
    res = {}
    levels = os.environ.get('COOP_LEVELS', '').split(',') if os.environ.get('COOP_LEVELS') else [None]
    cases = [('default', 0, None), ('lockstep', R.RT_LOCKSTEP_BVH | R.RT_NO_SPECULATE_BVH, None)] + [('joint' if lv is None else f'coop L{lv}', R.RT_LOCKSTEP_BVH | R.RT_JOINT_BVH, lv) for lv in levels]
    for tag, fl, lv in cases:
        if lv is not None: os.environ['RT_COOP_LEVELS'] = lv
        else: os.environ.pop('RT_COOP_LEVELS', None)
        ms = []
        for _ in range(3):
            out = R.render(b, cam, bg, W, H, spp, 50, flags=fl)
            ms.append(R.last_kernel_ms(b))
        res[tag] = (min(ms), out)
        extra = ''
        if tag.startswith('joint'):
            ts = R.last_traversal_stats(b)
            extra = f'  steps {ts["traversal_steps"]}, lanes/step {ts["traversal_lanes"] / max(1, ts["traversal_steps"]):.1f}, hand-overs {ts["leaf_steps"]}, redone+regives {ts["leaf_lanes"]}'
            import ctypes as C
            cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.c_void_p]; be.lib.rt_debug_section_cycles(b.h, cyc)
            if cyc[0]: extra += f'\n      per round: walking {cyc[1] / cyc[0]:.1f}, finished with a leaf {cyc[2] / cyc[0]:.1f}, spent helpers {cyc[3] / cyc[0]:.1f}, never used {cyc[4] / cyc[0]:.1f}, givers {cyc[5] / cyc[0]:.1f} (rounds {cyc[0]})'
        print(f'    {tag:9s} {min(ms):8.3f} ms  {W * H * spp / min(ms) / 1e3:8.1f} Msamples/s{extra}', flush=True)
    last = [t for t in res if t.startswith('joint')][-1]
    d = np.abs(res[last][1] - res['lockstep'][1]); fin = np.isfinite(d)
    print(f'    frame sums joint vs lockstep: max |diff| {d[fin].max():.3e}, non-finite pattern equal: {bool(np.array_equal(np.isfinite(res[last][1]), np.isfinite(res["lockstep"][1])))}')
