"""Details of the samples in which the GPU and the CPU oracle disagree for one random scene of tests/test_fuzz_gpu.py, and for the first of
them the hits of the path level by level on both sides (the kernel's from a -DRT_TRACE_PATH build: tools/mkab.sh trace "-DRT_TRACE_PATH"
"-DRT_TRACE_PATH"; uses the oracle: a developer probe, run by hand).   usage: python tests/sweeps/fuzz_probe.py seed [seed ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('RT_AMD_LIB', os.path.join(ROOT, 'raytracinginrust_amd/csrc/ab/trace.so'))
import numpy as np
import torch
from oracle import orc
from raytracinginrust_amd import _lib, render as R, scenes
from raytracinginrust_amd.api import CameraParams
from test_fuzz_gpu import _rand_scene, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
earth = scenes.load_earthmap()
W = H = 40; spp, depth = 8, 12
pbe.lib.rt_debug_trace_path.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong]
pbe.lib.rt_debug_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
obe.lib.orc_trace_path.restype = C.c_int
obe.lib.orc_trace_path.argtypes = [C.c_void_p, C.POINTER(CameraParams), C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
for seed in [int(x) for x in sys.argv[1:]]:
    ob, ocam, obg = _rand_scene(obe, seed, earth)
    pb, pcam, pbg = _rand_scene(pbe, seed, earth)
    print('seed', seed, 'tables', R.flatten(pb))
    ref, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=77 + seed, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, seed=77 + seed, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    fin = np.isfinite(rs_) & np.isfinite(gs)
    d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1) | (np.isnan(gs) != np.isnan(rs_)).any(axis=-1) | (np.isinf(gs) != np.isinf(rs_)).any(axis=-1)
    print('  bad samples', int(bad.sum()), 'gpu nonfinite', R.last_stats(pb)['nonfinite_samples'], 'oracle nonfinite', cnt['nonfinite'])
    for (y, x, k) in np.argwhere(bad)[:3]:
        print('    pixel', (int(x), int(y)), 'sample', int(k), 'gpu', gs[y, x, k], 'oracle', rs_[y, x, k])
        ot = np.zeros((depth + 1, 12)); bg = (C.c_double * 3)(*obg)
        n = obe.lib.orc_trace_path(ob.h, C.byref(ocam), bg, W, H, int(x), int(H - 1 - y), int(k), depth, 77 + seed, ot.ctypes.data, depth + 1)
        pbe.lib.rt_debug_trace_path(pb.h, int(y) * W + int(x), int(k))
        R.render(pb, pcam, pbg, W, H, spp, depth, seed=77 + seed, flags=R.RT_LOCKSTEP_BVH)
        gt = np.zeros((depth + 1, 16)); pbe.lib.rt_debug_get_trace(pb.h, gt.ctypes.data, depth + 1)
        pbe.lib.rt_debug_trace_path(pb.h, -1, -1)
        for lv in range(max(n, int((gt[:, 15] != 0).sum()) + 1)):
            o, g = ot[lv], gt[lv]
            same = (np.array_equal(o[:8], g[:8]) and np.array_equal(o[8:11], g[12:15])) or (not np.isfinite(o[0]) and g[15] == 0)
            print(f'      level {lv}: {"same" if same else "DIFFERENT"}')
            print(f'        oracle t {o[0]!r} p {o[1:4]} n {o[4:7]} front {o[7]} dir in {[float(v) for v in o[8:11]]!r}')
            print(f'        gpu    t {g[0]!r} p {g[1:4]} n {g[4:7]} front {g[7]}  object {int(g[8])} prim kind {int(g[9])} index {int(g[10])} material {int(g[11])} dir in {[float(v) for v in g[12:15]]!r}' if g[15] else '        gpu    (no hit recorded)')
            if not same: break
