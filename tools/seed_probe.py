"""For seeds a fuzz sweep flagged (tests/sweeps/fuzz_sweep.py): the same random scene rendered with TWO builds of the library in one process,
every sample compared bit for bit.  A flagged seed that gives the same words with an older build is the documented device-vs-oracle
class (a last-ulp libm difference that sends a path down another branch, or a throughput that overflows: DESIGN.md §6 (a) / (b)), not
something a kernel change introduced.   usage: python tools/seed_probe.py old=path/to/old.so new=path/to/new.so seed [seed ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, scenes
from test_fuzz_gpu import _rand_scene
libs = [a.split('=', 1) for a in sys.argv[1:] if '=' in a]
seeds = [int(a) for a in sys.argv[1:] if '=' not in a]
assert len(libs) == 2 and seeds, __doc__
backends = [(n, _lib.load_path(os.path.abspath(p))) for n, p in libs]
earth = scenes.load_earthmap()
W = H = 40; spp, depth = 8, 12                 # the sweep's frame
total = 0
for seed in seeds:
    res = []
    for name, be in backends:
        pb, pcam, pbg = _rand_scene(be, seed, earth)
        out = np.zeros((H, W, 3)); smp = np.zeros((H, W, spp, 3))
        assert be.lib.rt_render_samples(pb.h, C.byref(pcam), (C.c_double * 3)(*pbg), W, H, spp, depth, 77 + seed, 0, out.ctypes.data, smp.ctypes.data) == 0
        res.append(smp)
    n = int((res[0].view(np.uint64) != res[1].view(np.uint64)).sum()); total += n
    print(f'{seed}: {backends[0][0]} vs {backends[1][0]}: {n} differing 64-bit words')
print(f'total over {len(seeds)} seeds: {total} differing words')
