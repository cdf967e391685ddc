"""One-off wide fuzz of round 6's two scene families (tests/test_fuzz_gpu.py): BVHs whose children are any Hittable (lists, wrapped
objects, media, nested BVHs: the F_NESTED instantiation) and rooms of parallel rect pairs (the lean kernel) — GPU vs CPU oracle, per
sample.   usage: python tests/sweeps/nested_scene_sweep.py [first_seed] [n_nested] [n_rooms]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from oracle import orc
from raytracinginrust_amd import _lib, render as R
from test_fuzz_gpu import _rand_nested_scene, _rand_room_scene, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
counts = {"nested": int(sys.argv[2]) if len(sys.argv) > 2 else 300, "rooms": int(sys.argv[3]) if len(sys.argv) > 3 else 300}
W = H = 40; spp, depth = 8, 12
for name, make, want_feats in (("nested", _rand_nested_scene, 639), ("rooms", _rand_room_scene, 0)):
    worst = 0.0; n_bad = 0; n_samples = 0; failures = []; plain = 0
    for seed in range(first, first + counts[name]):
        ob, ocam, obg = make(obe, seed)
        pb, pcam, pbg = make(pbe, seed)
        ref, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=53 + seed, want_samples=True, want_counters=True)
        got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, seed=53 + seed, want_samples=True)
        nan_ok = np.array_equal(np.isnan(gs), np.isnan(rs_)) and np.array_equal(np.isinf(gs), np.isinf(rs_))
        fin = np.isfinite(rs_) & np.isfinite(gs)
        d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
        bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
        n_bad += int(bad.sum()); n_samples += bad.size
        keep = ~np.repeat(bad[..., None], 3, -1).reshape(d.shape)
        worst = max(worst, float(d[keep].max()))
        plain += R.last_loop_info(pb)['feats'] != want_feats          # (a nested scene whose random children were all bare primitives runs a plain kernel)
        if not nan_ok or bad.sum() > 2 or R.last_stats(pb)['nonfinite_samples'] != cnt['nonfinite']:
            failures.append((seed, nan_ok, int(bad.sum()), R.last_loop_info(pb)['feats']))
        if (seed - first + 1) % 500 == 0:
            print(f'  ... {name}: {seed - first + 1} scenes, {n_samples} samples, {n_bad} diverged, failing seeds so far {failures}', flush=True)
    print(f'{name} scenes, seeds {first}..{first + counts[name] - 1}: {n_samples} samples, {n_bad} diverged (path took another branch after a last-ulp '
          f'difference), worst |gpu - oracle| among the rest {worst:.3e}; {plain} scenes ran another instantiation than {want_feats}; failing seeds: {failures}', flush=True)
