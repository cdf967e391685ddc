import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import ctypes as C
from raytracinginrust_amd import _lib, render as R, scenes
from test_fuzz_gpu import _rand_scene
earth = scenes.load_earthmap()
libs = {'base': _lib.load_path(os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/base.so')), 'new': _lib.load_path(os.path.join(ROOT, 'raytracinginrust_amd/csrc/librt_amd.so'))}
W = H = 40; spp, depth = 8, 12
tot = 0
for seed in [int(x) for x in sys.argv[1:]]:
    res = {}
    for name, be in libs.items():
        pb, pcam, pbg = _rand_scene(be, seed, earth)
        out = np.zeros((H, W, 3)); smp = np.zeros((H, W, spp, 3))
        assert be.lib.rt_render_samples(pb.h, C.byref(pcam), (C.c_double * 3)(*pbg), W, H, spp, depth, 77 + seed, 0, out.ctypes.data, smp.ctypes.data) == 0
        res[name] = smp
    n = int((res['base'].view(np.uint64) != res['new'].view(np.uint64)).sum()); tot += n
    print(seed, 'round-4 library vs this build, differing words:', n)
print('total differing words over the flagged seeds:', tot)
