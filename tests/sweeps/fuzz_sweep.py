"""One-off wide fuzz: tests/test_fuzz_gpu.py's random scenes for many more seeds (GPU vs CPU oracle, per sample).
usage: python tests/sweeps/fuzz_sweep.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from PIL import Image
from oracle import orc
from raytracinginrust_amd import _lib, render as R, scenes
from test_fuzz_gpu import _rand_scene, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
earth = scenes.load_earthmap()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = H = 40; spp, depth = 8, 12
worst = 0.0; n_bad_total = 0; n_samples = 0; failures = []
for seed in range(first, first + n):
    ob, ocam, obg = _rand_scene(obe, seed, earth)
    pb, pcam, pbg = _rand_scene(pbe, seed, earth)
    ref, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=77 + seed, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, seed=77 + seed, want_samples=True)
    nan_ok = np.array_equal(np.isnan(gs), np.isnan(rs_)) and np.array_equal(np.isinf(gs), np.isinf(rs_))
    fin = np.isfinite(rs_) & np.isfinite(gs)
    d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
    n_bad_total += int(bad.sum()); n_samples += bad.size
    ok_d = d[~np.repeat(bad[..., None], 3, -1).reshape(d.shape)] if bad.any() else d
    worst = max(worst, float(ok_d.max()))
    if not nan_ok or bad.sum() > 2 or R.last_stats(pb)['nonfinite_samples'] != cnt['nonfinite']:
        failures.append((seed, nan_ok, int(bad.sum())))
print(f'seeds {first}..{first + n - 1}: {n_samples} samples, {n_bad_total} diverged (path took another branch after a last-ulp difference), '
      f'worst |gpu - oracle| among the rest {worst:.3e}; failing seeds: {failures}')

# mesh rooms (the persistent-traversal kernel): both loop shapes against each other (bit for bit) and against the oracle
from test_parity_gpu import _mesh_room
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 100
bad_bits = 0; worst_m = 0.0; div_m = 0
for seed in range(first, first + nm):
    b, cam, bg = _mesh_room(pbe, seed)
    _, lock = R.render(b, cam, bg, W, H, spp, 16, seed=5 + seed, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    _, pers = R.render(b, cam, bg, W, H, spp, 16, seed=5 + seed, flags=R.RT_PERSISTENT_BVH, want_samples=True)
    bad_bits += int((lock.view(np.uint64) != pers.view(np.uint64)).sum())
    ob, ocam, obg = _mesh_room(obe, seed)
    _, ref = orc.render(ob, ocam, obg, W, H, spp, 16, seed=5 + seed, want_samples=True)
    fin = np.isfinite(ref) & np.isfinite(pers)
    d = np.abs(np.where(fin, pers, 0.0) - np.where(fin, ref, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, ref, 0.0)))).any(axis=-1)
    div_m += int(bad.sum()); worst_m = max(worst_m, float(d[~bad].max()))
print(f'mesh rooms {first}..{first + nm - 1}: persistent vs lock-step differing words {bad_bits}; vs oracle {div_m} diverged samples, worst of the rest {worst_m:.3e}')
