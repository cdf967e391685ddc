"""One-off wide fuzz of the SIMPLE room form in mesh scenes (round 6: walls that are faces of one box and stand next to each other in the list of a
scene with triangle-mesh BVHs — the teapot room's shape; rt_flatten.cpp form_room, rt_kernel.hip RoomSite): tests/test_parity_gpu.py's walled
mesh rooms for many seeds — per sample against the oracle in both loop shapes, persistent == lock-step and room == no room (RT_NO_ROOM) word for word.
usage: python tests/sweeps/walled_mesh_room_sweep.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
from oracle import orc
from raytracinginrust_amd import _lib, render as R
from test_parity_gpu import _walled_mesh_room, SAMPLE_RTOL
pbe, obe = _lib.load(), orc.load()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 300
W, H, spp, depth = 48, 40, 8, 20
n_bad = n_samples = n_rooms = words_loop = words_room = 0; worst = 0.0; failures = []
for seed in range(first, first + n):
    ob, ocam, obg = _walled_mesh_room(obe, seed)
    pb, pcam, pbg = _walled_mesh_room(pbe, seed)
    _, rs_, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, seed=77 + seed, want_samples=True, want_counters=True)
    _, gp = R.render(pb, pcam, pbg, W, H, spp, depth, seed=77 + seed, flags=R.RT_PERSISTENT_BVH, want_samples=True)
    _, gl = R.render(pb, pcam, pbg, W, H, spp, depth, seed=77 + seed, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    has_room = any(o["is_cube"] & 2 for o in R.debug_objects(pb)); n_rooms += int(has_room)
    os.environ["RT_NO_ROOM"] = "1"
    qb, qcam, qbg = _walled_mesh_room(pbe, seed)
    _, g0 = R.render(qb, qcam, qbg, W, H, spp, depth, seed=77 + seed, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    del os.environ["RT_NO_ROOM"]
    wl = int((gp.view(np.uint64) != gl.view(np.uint64)).sum()); wr = int((gl.view(np.uint64) != g0.view(np.uint64)).sum())
    words_loop += wl; words_room += wr
    nan_ok = np.array_equal(np.isnan(gl), np.isnan(rs_)) and np.array_equal(np.isinf(gl), np.isinf(rs_))
    fin = np.isfinite(rs_) & np.isfinite(gl)
    d = np.abs(np.where(fin, gl, 0.0) - np.where(fin, rs_, 0.0))
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs_, 0.0)))).any(axis=-1)
    n_bad += int(bad.sum()); n_samples += bad.size
    keep = ~np.repeat(bad[..., None], 3, -1).reshape(d.shape)
    worst = max(worst, float(d[keep].max()))
    if not nan_ok or bad.sum() > 2 or wl or wr or not has_room or R.last_stats(pb)['nonfinite_samples'] != cnt['nonfinite']:
        failures.append((seed, nan_ok, int(bad.sum()), wl, wr, has_room))
print(f'walled mesh rooms, seeds {first}..{first + n - 1}: {n_rooms} of {n} scenes form a room, {n_samples} samples, {n_bad} diverged from the oracle, worst |gpu - oracle| among the rest {worst:.3e}; '
      f'differing 64-bit words persistent vs lock-step {words_loop}, room vs plain list {words_room}; failing seeds: {failures}', flush=True)
