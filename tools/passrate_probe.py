"""EXPERIMENT (round 6, verdict item 2c): contraction of the filter tree guided by MEASURED pass rates instead of box areas.
Needs the counting build of the library (never shipped; the hooks are compiled in by -DRT_COUNT_NODES only):

    tools/mkcount.sh          ->  csrc/abx/count.so   (kernels' box steps count visits / passes per node; the flattener reads RT_COLLAPSE_SET)

Per scene: (1) one frame with the counters on (RT_NODE_COUNTS=1) and NO contraction (RT_COLLAPSE_TAU=2) -> visits and passes of every
node of the reference's tree for this view; (2) per threshold p*: the set of inner non-root nodes whose pass rate exceeds p* is written
to a file, RT_COLLAPSE_SET names it, the scene is flattened again and a frame is timed; (3) the same with the shipped area rule
(tau = 0.75) and with no contraction.  All on the one library, counters off while timing.  Every variant's samples are the same bit
for bit (the filter tree only decides what is skipped) — checked.      usage: python tools/passrate_probe.py [scene ...]"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/count.so')
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R, scenes
be = _lib.load()
lib = be.lib
lib.rt_debug_node_counts.restype = C.c_int
lib.rt_debug_node_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
CFG = {"final": (800, 800, 64, 50), "random": (800, 800, 256, 50), "teapot": (800, 800, 64, 50)}


def build(name):
    if name == "final": return scenes.final_scene(be, *scenes.load_earthmap())
    if name == "random": return scenes.random_scene(be, aspect_ratio=1.0)
    return scenes.cornell_test(be, scenes.asset_path('teapot.obj'))


def timed(name, env, rounds=4, want=None):
    for k, v in env.items(): os.environ[k] = v
    b, cam, bg = build(name)
    W, H, spp, depth = CFG[name]
    ms = []
    img = None
    for _ in range(rounds):
        img = R.render(b, cam, bg, W, H, spp, depth)
        ms.append(R.last_kernel_ms(b))
    for k in env: os.environ.pop(k, None)
    if want is not None:
        assert np.all(np.abs(img - want) <= 1e-12 * (spp + np.abs(want))), "a filter tree changed the samples"
    return min(ms[1:]), img


def host_estimate(L, eb, roots, n_rays=3000, seed=1, cam21=None, cam_share=0.5):
    """Pass rates WITHOUT a GPU launch or a view: rays from random points of random leaf boxes in uniform directions, the ordered walk of
    bvh.rs:77-91 simulated on the host with every leaf box standing in for its primitive (a leaf whose box is entered is a hit at the
    entry distance), visits / passes counted per node.  What a flattener could do by itself."""
    rs = np.random.RandomState(seed)
    n = len(L)
    LEAF, DONE = 1 << 31, 0xFFFFFFFF
    leaves = np.flatnonzero((L[:, 0] & LEAF) != 0)
    visits = np.zeros(n); passes = np.zeros(n)
    mn, mx = eb[:, :3], eb[:, 3:]
    for k in range(n_rays):
        if cam21 is not None and k < cam_share * n_rays:             # primary rays of the view (Camera::get_ray without the lens, camera.rs:51-59)
            c = np.array(cam21)
            o = c[0:3].copy()
            d = c[3:6] + rs.rand() * c[6:9] + rs.rand() * c[9:12] - o
        else:
            lf = leaves[rs.randint(len(leaves))]
            ext = np.minimum(mx[lf] - mn[lf], 1e4)                      # (a giant ground sphere: stay near the scene)
            o = (mn[lf] + mx[lf]) * 0.5 + (rs.rand(3) - 0.5) * ext
            d = rs.normal(size=3); d /= np.linalg.norm(d)
            o = o + d * (0.51 * np.linalg.norm(ext))                    # leave the leaf it starts on
        inv = 1.0 / d
        for root in roots:
            closest = np.inf
            i = int(root)
            while i != DONE:
                t0 = (mn[i] - o) * inv; t1 = (mx[i] - o) * inv
                t_in = max(1e-5, np.minimum(t0, t1).max()); t_out = min(closest, np.maximum(t0, t1).min())
                visits[i] += 1
                ok = t_out > t_in
                if ok: passes[i] += 1
                if L[i, 0] & LEAF:
                    if ok: closest = min(closest, t_in)
                    i = int(L[i, 3])
                else:
                    i = int(L[i, 2]) if ok else int(L[i, 3])
    return visits, passes


for name in (sys.argv[1:] or ["final", "random", "teapot"]):
    W, H, spp, depth = CFG[name]
    # (1) counts on the uncontracted tree
    os.environ['RT_NODE_COUNTS'] = '1'; os.environ['RT_COLLAPSE_TAU'] = '2'
    b, cam, bg = build(name)
    n = R.flatten(b)["bvh_nodes"]
    R.render(b, cam, bg, W, H, max(4, spp // 8), depth)
    cnt = np.zeros((n, 2), np.uint64)
    assert lib.rt_debug_node_counts(b.h, cnt.ctypes.data, n) == n
    os.environ.pop('RT_NODE_COUNTS'); os.environ.pop('RT_COLLAPSE_TAU')
    links = (C.c_uint32 * (4 * n))(); roots = (C.c_uint32 * 16)(); nr = C.c_uint32(0)
    lib.rt_debug_bvh_links(b.h, links, n, roots, 16, C.byref(nr))
    L = np.frombuffer(links, np.uint32).reshape(n, 4)
    inner = (L[:, 0] & (1 << 31)) == 0
    is_root = np.zeros(n, bool); is_root[list(roots)[:nr.value]] = True
    visits, passes = cnt[:, 0].astype(np.float64), cnt[:, 1].astype(np.float64)
    rate = np.where(visits > 0, passes / np.maximum(visits, 1), 0.0)
    print(f"== {name}: {n} nodes, {int(inner.sum())} inner; box tests per frame {visits.sum():.3e}, passed {passes.sum() / visits.sum():.3f}; "
          f"inner non-root nodes with pass rate > 0.5 / 0.7 / 0.8 / 0.9 / 0.95: " + " / ".join(str(int(((rate > t) & inner & ~is_root).sum())) for t in (0.5, 0.7, 0.8, 0.9, 0.95)))
    base_ms, want = timed(name, {"RT_COLLAPSE_TAU": "2"})
    print(f"   no contraction            {base_ms:9.3f} ms")
    area_ms, _ = timed(name, {}, want=want)
    print(f"   area rule, tau = 0.75     {area_ms:9.3f} ms   ({(base_ms / area_ms - 1) * 100:+.1f} % vs none)")
    eb = np.zeros((n, 6)); fm = C.c_float(0)
    lib.rt_debug_filter_nodes.restype = C.c_int
    lib.rt_debug_filter_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    lib.rt_debug_filter_nodes(b.h, None, None, eb.ctypes.data, n, C.byref(fm))
    from raytracinginrust_amd.api import camera_fields
    cam21 = camera_fields(be, cam)
    for label, kw in (("no view", {}), ("half of the rays are the view's primary rays", {"cam21": cam21}), ("primary rays only", {"cam21": cam21, "cam_share": 1.0})):
        hv, hp = host_estimate(L, eb, list(roots)[:nr.value], **kw)
        hrate = np.where(hv > 0, hp / np.maximum(hv, 1), 0.0)
        both = inner & ~is_root & (visits > 0) & (hv > 0)
        print(f"   host estimate (3000 rays, {label}): correlation with the measured pass rates {np.corrcoef(rate[both], hrate[both])[0, 1]:.3f}")
        for thr in (0.7, 0.8, 0.9):
            ids = np.flatnonzero((hrate > thr) & inner & ~is_root & (hv > 0))
            f = tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False); f.write("\n".join(str(int(i)) for i in ids)); f.close()
            ms, _ = timed(name, {"RT_COLLAPSE_SET": f.name}, want=want)
            os.unlink(f.name)
            print(f"      > {thr:4.2f} ({len(ids):4d} nodes) {ms:9.3f} ms   ({(area_ms / ms - 1) * 100:+.1f} % vs the area rule)")
    for thr in (0.6, 0.7, 0.8, 0.85, 0.9, 0.95):
        ids = np.flatnonzero((rate > thr) & inner & ~is_root & (visits > 0))
        f = tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False); f.write("\n".join(str(int(i)) for i in ids)); f.close()
        ms, _ = timed(name, {"RT_COLLAPSE_SET": f.name}, want=want)
        os.unlink(f.name)
        print(f"   pass rate > {thr:4.2f} ({len(ids):4d} nodes) {ms:9.3f} ms   ({(area_ms / ms - 1) * 100:+.1f} % vs the area rule)")
