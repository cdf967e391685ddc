"""One-off fuzz of the view-tuned filter tree (round 6): worlds that are ONE bare BVH — scenes.random_scene with other seeds — from random
cameras: every sample of the tuned render against the CPU oracle's, and against the same scene rendered with the area rule's tree
(RT_NO_FILTER_TUNING) bit for bit.   usage: python tests/sweeps/one_bvh_sweep.py [first_seed] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from oracle import orc
from raytracinginrust_amd import _lib, render as R, scenes
from raytracinginrust_amd.api import Camera
pbe, obe = _lib.load(), orc.load()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H, spp, depth = 64, 48, 8, 8
n_samples = n_bad = words = 0; worst = 0.0; failures = []
for seed in range(first, first + n):
    rs = np.random.RandomState(seed)
    frm = (float(rs.uniform(-14, 14)), float(rs.uniform(0.5, 9)), float(rs.uniform(-14, 14)))
    cam_args = (frm, (float(rs.uniform(-3, 3)), float(rs.uniform(0, 1)), float(rs.uniform(-3, 3))), (0.0, 1.0, 0.0), float(rs.uniform(15, 60)), W / H,
                float(rs.choice([0.0, 0.1])), 10.0, 0.0, 1.0)
    pb, _, bg = scenes.random_scene(pbe, seed=seed, aspect_ratio=W / H)
    ob, _, _ = scenes.random_scene(obe, seed=seed, aspect_ratio=W / H)
    cam = Camera(*cam_args)
    _, gs = R.render(pb, cam, bg, W, H, spp, depth, seed=9 + seed, want_samples=True)
    _, rs_ = orc.render(ob, cam, bg, W, H, spp, depth, seed=9 + seed, want_samples=True)
    os.environ['RT_NO_FILTER_TUNING'] = '1'
    pb2, _, _ = scenes.random_scene(pbe, seed=seed, aspect_ratio=W / H)
    _, plain = R.render(pb2, cam, bg, W, H, spp, depth, seed=9 + seed, want_samples=True)
    os.environ.pop('RT_NO_FILTER_TUNING')
    words += int((plain.view(np.uint64) != gs.view(np.uint64)).sum())
    fin = np.isfinite(rs_) & np.isfinite(gs)
    d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > 1e-9 * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
    n_bad += int(bad.sum()); n_samples += bad.size
    worst = max(worst, float(d[~np.repeat(bad[..., None], 3, -1).reshape(d.shape)].max()))
    if bad.sum() > 2 or not np.array_equal(np.isfinite(gs), np.isfinite(rs_)) or R.last_loop_info(pb)['feats'] != 2111:
        failures.append((seed, int(bad.sum())))
print(f'one-BVH worlds, seeds {first}..{first + n - 1}: {n_samples} samples, {n_bad} diverged from the oracle, worst of the rest {worst:.3e}; '
      f'tuned vs area-rule tree: {words} differing words; failing seeds: {failures}', flush=True)
