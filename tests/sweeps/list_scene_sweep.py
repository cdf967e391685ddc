"""One-off wide fuzz of the lean list-scene kernels: tests/test_fuzz_gpu.py's random list scenes for many more seeds (GPU vs CPU
oracle, per sample).  usage: python tests/sweeps/list_scene_sweep.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from oracle import orc
from raytracinginrust_amd import _lib, render as R
from test_fuzz_gpu import _rand_list_scene, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 500
W = H = 40; spp, depth = 8, 12
worst = 0.0; n_bad = 0; n_samples = 0; failures = []
for seed in range(first, first + n):
    ob, ocam, obg = _rand_list_scene(obe, seed)
    pb, pcam, pbg = _rand_list_scene(pbe, seed)
    ref, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=31 + seed, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, seed=31 + seed, want_samples=True)
    nan_ok = np.array_equal(np.isnan(gs), np.isnan(rs_)) and np.array_equal(np.isinf(gs), np.isinf(rs_))
    fin = np.isfinite(rs_) & np.isfinite(gs)
    d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
    n_bad += int(bad.sum()); n_samples += bad.size
    keep = ~np.repeat(bad[..., None], 3, -1).reshape(d.shape)
    worst = max(worst, float(d[keep].max()))
    if not nan_ok or bad.sum() > 2 or R.last_stats(pb)['nonfinite_samples'] != cnt['nonfinite'] or R.last_launch_info(pb)['threads'] != 256:
        failures.append((seed, nan_ok, int(bad.sum())))
print(f'list scenes, seeds {first}..{first + n - 1}: {n_samples} samples, {n_bad} diverged (path took another branch after a last-ulp difference), '
      f'worst |gpu - oracle| among the rest {worst:.3e}; failing seeds: {failures}')
