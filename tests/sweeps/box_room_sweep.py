"""One-off wide fuzz of the ROOM form (round 6: bare AARects that are exact faces of one box, one object through the Cube fast path;
rt_flatten.cpp form_room, rt_kernel.hip cube_fast): tests/test_fuzz_gpu.py's random box rooms — walls in any list order, patches and
second walls that tie with them bit for bit before / between / after them, cameras inside, outside, on planes, in corners — for many more
seeds, GPU vs CPU oracle per sample, and the same scenes with RT_NO_ROOM (the list as the reference has it) word for word.
usage: python tests/sweeps/box_room_sweep.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from oracle import orc
from raytracinginrust_amd import _lib, render as R
from test_fuzz_gpu import _rand_box_room_scene, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 500
W = H = 40; spp, depth = 8, 12
worst = 0.0; n_bad = 0; n_samples = 0; failures = []; n_rooms = 0; n_words = 0
for seed in range(first, first + n):
    ob, ocam, obg, _ = _rand_box_room_scene(obe, seed)
    pb, pcam, pbg, n_faces = _rand_box_room_scene(pbe, seed)
    ref, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=31 + seed, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, seed=31 + seed, want_samples=True)
    has_room = any(o["is_cube"] & 2 for o in R.debug_objects(pb))
    n_rooms += int(has_room)
    os.environ["RT_NO_ROOM"] = "1"
    qb, qcam, qbg, _ = _rand_box_room_scene(pbe, seed)
    _, gs0 = R.render(qb, qcam, qbg, W, H, spp, depth, seed=31 + seed, want_samples=True)
    assert not any(o["is_cube"] & 2 for o in R.debug_objects(qb))
    del os.environ["RT_NO_ROOM"]
    words = int((gs.view(np.uint64) != gs0.view(np.uint64)).sum())
    n_words += words
    nan_ok = np.array_equal(np.isnan(gs), np.isnan(rs_)) and np.array_equal(np.isinf(gs), np.isinf(rs_))
    fin = np.isfinite(rs_) & np.isfinite(gs)
    d = np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
    n_bad += int(bad.sum()); n_samples += bad.size
    keep = ~np.repeat(bad[..., None], 3, -1).reshape(d.shape)
    worst = max(worst, float(d[keep].max()))
    if not nan_ok or bad.sum() > 2 or words or R.last_stats(pb)['nonfinite_samples'] != cnt['nonfinite'] or R.last_launch_info(pb)['threads'] != 256:
        failures.append((seed, nan_ok, int(bad.sum()), words, has_room, n_faces))
print(f'box rooms, seeds {first}..{first + n - 1}: {n_rooms} of {n} scenes form a room, {n_samples} samples, {n_bad} diverged from the oracle, '
      f'worst |gpu - oracle| among the rest {worst:.3e}; {n_words} differing 64-bit words between the room form and the plain list; failing seeds: {failures}', flush=True)
