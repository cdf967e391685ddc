import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import build_scene
from raytracinginrust_amd import _lib, render as R, scenes
pbe = _lib.load(); earth = scenes.load_earthmap()
name = sys.argv[1]
b, cam, bg = build_scene(name, pbe, earth)
W, H, spp, depth = 96, 54, 8, 30
ref, rs = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_NO_DEFER_BVH, want_samples=True)
print('ref ok', flush=True)
for below, stop in (("48", "20"), ("65", "64"), ("0", "20"), ("7", "1"), ("65", "33"), ("30", "12")):
    os.environ["RT_DEFER_DENSE"] = below; os.environ["RT_DEFER_STOP"] = stop
    t = time.time()
    print('start', below, stop, flush=True)
    got, gs = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_DEFER_BVH, want_samples=True)
    print('done', below, stop, np.array_equal(rs.view(np.uint64), gs.view(np.uint64)), R.last_traversal_stats(b), f'{time.time()-t:.2f}s', flush=True)
