"""Host-side logic of the product library (no GPU needed): the C-ABI loads and exports what include/rt_amd.h
declares, the seeded stream / Camera::new / format_color / PPM emitter agree with the oracle bit for bit,
scenes flatten to the expected device tables, and error paths return errors instead of panicking."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, build_scene, golden_path
from oracle import orc
from raytracinginrust_amd import _lib, render as R, scenes
from raytracinginrust_amd.api import Axis, Camera, Plane, Rng, SceneBuilder, SceneError, camera_fields, format_color


def test_library_exports_every_declared_symbol(pbe):
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 40
    missing = [n for n in names if not hasattr(pbe.lib, n)]
    assert missing == []


def test_rng_streams_match_oracle_bitwise(obe, pbe):
    for seed, stream in [(0x5EED, 0), (0, 1), (2 ** 64 - 1, 16), (123456789, 7)]:
        a, b = Rng(obe, seed, stream), Rng(pbe, seed, stream)
        for k in range(300):
            m = k % 5
            if m == 0:
                assert a.gen_f64() == b.gen_f64()
            elif m == 1:
                assert a.gen_range(-1.0, 1.0) == b.gen_range(-1.0, 1.0)
            elif m == 2:
                assert a.gen_bool() == b.gen_bool()
            elif m == 3:
                assert a.gen_index(k + 1) == b.gen_index(k + 1)
            else:
                assert a.gen_range(213.0, 343.0) == b.gen_range(213.0, 343.0)
    sa, sb = (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    for pix, s in [(0, 0), (1, 0), (0, 1), (639999, 1023), (8294399, 8191), (2 ** 31 - 1, 2 ** 32 - 1)]:
        obe.fn("rng_path")(0x5EED, pix, s, sa)
        pbe.fn("rng_path")(0x5EED, pix, s, sb)
        assert list(sa) == list(sb)


CAMERAS = [
    ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, 16.0 / 9.0, 0.1),          # main.rs:630-635
    ((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 40.0, 1.0, 0.05),      # main.rs:700-705
    ((199.0, 439.0, -200.0), (278.0, 375.0, 258.0), 30.0, 16.0 / 9.0, 0.01),   # main.rs:728-733
    ((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 40.0, 1.0, 0.01),      # main.rs:742-747
]


@pytest.mark.parametrize("lf,la,vfov,aspect,ap", CAMERAS)
def test_camera_new_matches_oracle_bitwise(obe, pbe, lf, la, vfov, aspect, ap):
    cam = Camera(lf, la, (0.0, 1.0, 0.0), vfov, aspect, ap, 10.0, 0.0, 1.0)
    assert camera_fields(obe, cam) == camera_fields(pbe, cam)


def test_format_color_matches_oracle(obe, pbe):
    rs = np.random.RandomState(0)
    vals = list(rs.uniform(0, 40, size=(200, 3))) + [
        (float("nan"), float("inf"), -1.0), (0.0, -0.0, 1e-320), (16.0, 15.999, 16.001), (1e300, 1e-300, 3.99)]
    for v in vals:
        for spp in (1, 16, 1024):
            assert format_color(obe, v, spp) == format_color(pbe, v, spp)
    assert format_color(pbe, (float("nan"), float("inf"), 4.0), 16) == (0, 255, 128)


def test_write_ppm_layout(tmp_path, pbe):
    """src/main.rs:767-769,832: P3 header, one `r g b` line per pixel, rows top to bottom"""
    img = np.zeros((2, 3, 3))
    img[0, 0] = (16.0, 0.0, 0.0)
    img[1, 2] = (0.0, 4.0, float("nan"))
    p = str(tmp_path / "o.ppm")
    R.write_ppm(p, img, 16)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"]
    assert lines[3] == "255 0 0" and lines[3 + 5] == "0 128 0" and len([l for l in lines if l]) == 3 + 6


def test_flatten_tables(pbe, earth):
    c = R.flatten(build_scene("cornell", pbe)[0])
    # 6 wall/light rects + 2 x 6 cube faces; the five walls are faces of one box: one ROOM object where the last of them stood (round 6;
    # its run of rect records = copies of the five + two records that carry the box): [FlipNormal(light)] [room] [box] [box]
    # (the object table also keeps the list as the reference has it, five objects, behind the four: what a wave searches when one of its
    # rays could produce a NaN plane distance — rt_kernel.hip world_hit)
    assert (c["objects"], c["ops"], c["rects"], c["lights"], c["bvh_nodes"]) == (4 + 5, 5, 18 + 5 + 2, 1, 0)
    r = R.flatten(build_scene("random", pbe)[0])
    assert r["spheres"] + r["moving_spheres"] == 533 and r["bvh_nodes"] == 2 * 533 - 1 and r["lights"] == 0
    f = R.flatten(build_scene("final", pbe, earth)[0])
    assert f["rects"] == 400 * 6 + 1 and f["spheres"] == 1000 + 6 - 1 + 1 and f["moving_spheres"] == 1
    assert f["bvh_nodes"] == (2 * 400 - 1) + (2 * 1000 - 1) and f["media"] == 2 and f["perlins"] == 1 and f["objects"] == 11 - 3
    t = R.flatten(build_scene("teapot", pbe)[0])
    # (the teapot room's five walls stand next to each other in the list: one room object in place — the simple form, no second list —, its
    # run of rect records = copies of the five + two records that carry the box; then the light, then the mesh's BVH)
    assert t["triangles"] == 1024 and t["bvh_nodes"] == 2047 and t["rects"] == 6 + 5 + 2 and t["objects"] == 3


def _objects(pbe, b):
    pbe.lib.rt_debug_objects.restype = C.c_int
    pbe.lib.rt_debug_objects.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    n_top = C.c_uint32(0)
    n = pbe.lib.rt_debug_objects(b.h, None, 0, C.byref(n_top))
    assert n >= 0
    out = np.zeros((max(n, 1), 8), np.uint32)
    assert pbe.lib.rt_debug_objects(b.h, out.ctypes.data, n, C.byref(n_top)) == n
    return out[:n], int(n_top.value)


def test_object_table_of_the_cornell_box(pbe, monkeypatch):
    """rt_debug_objects: [FlipNormal(light)] [room: the five walls] [box] [box] — and, with RT_NO_ROOM, the list as the reference has it:
    [green, red] [FlipNormal(light)] [floor, ceiling, back] [box] [box] (runs of bare rects merge; a Cube and a FlipNormal-only chain are marked)."""
    ob, n_top = _objects(pbe, build_scene("cornell", pbe)[0])
    assert n_top == 4 and len(ob) == 4 + 5
    reference_list = [(0, 2, 0, 0, 0), (0, 1, 1, 0, 0x10000), (0, 3, 0, 0, 0), (0, 6, 2, 1, 0), (0, 6, 2, 1, 0)]
    assert [tuple(int(x) for x in o[[0, 2, 4, 6, 7]]) for o in ob[4:]] == reference_list      # behind the world list: the list as the reference has it (world_hit's NaN-proof path)
    ob = ob[:4]
    assert [tuple(int(x) for x in o[[0, 2, 4, 7]]) for o in ob] == [(0, 1, 1, 0x10000), (0, 5, 0, 0), (0, 6, 2, 0), (0, 6, 2, 0)]      # 0x10000: every wrapper of the light is a FlipNormal (the lean kernel tests the path's own ray)
    room = ob[1]
    assert int(room[1]) == 18 and int(room[6]) & 0xFF == 2 and int(ob[0][6]) == 0 and int(ob[2][6]) == int(ob[3][6]) == 1
    # faces in cube.rs:17-24 order (z max, z min, y max, y min, x max, x min) -> place in the run, which keeps main.rs:291-296's order
    # [x = 555 green, x = 0 red, y = 0 floor, y = 555 ceiling, z = 555 back]; 7: the open front
    assert [(int(room[6]) >> (8 + 3 * f)) & 7 for f in range(6)] == [4, 7, 3, 2, 0, 1]
    # the tie rule: the light (new index 0) stood after the first two walls and before the other three (whose entry is the room's own index)
    assert [(int(room[3]) >> (5 * j)) & 31 for j in range(5)] == [0, 0, 1, 1, 1]
    monkeypatch.setenv("RT_NO_ROOM", "1")
    ob, n_top = _objects(pbe, build_scene("cornell", pbe)[0])
    assert n_top == len(ob) == 5
    assert [tuple(int(x) for x in o[[0, 2, 4, 6, 7]]) for o in ob] == reference_list


def _room_list(pbe, spec, lights=False):
    """A list scene from a compact spec: ("w", plane, hi) a wall of the box (0,0,0)-(4,6,8) on that face, ("p", plane, hi) a smaller
    patch in a face's plane, ("r", ...) a rect that is no face at all, ("f", item) a FlipNormal around an item, ("c",) a rotated Cube."""
    b = SceneBuilder(pbe)
    m = b.Lambertian(b.ConstantTexture((0.5, 0.5, 0.5)))
    mn, mx = (0.0, 0.0, 0.0), (4.0, 6.0, 8.0)
    ax = {Plane.XY: (2, 0, 1), Plane.XZ: (1, 0, 2), Plane.YZ: (0, 1, 2)}

    def make(it):
        if it[0] == "w":
            k, a_, b_ = ax[it[1]]
            return b.AARect(it[1], mn[a_], mx[a_], mn[b_], mx[b_], mx[k] if it[2] else mn[k], m)
        if it[0] == "p":
            k, a_, b_ = ax[it[1]]
            return b.AARect(it[1], mn[a_] + 1.0, mx[a_] - 1.0, mn[b_] + 1.0, mx[b_] - 1.0, mx[k] if it[2] else mn[k], m)
        if it[0] == "r":
            return b.AARect(Plane.XY, -1.0, 1.0, -1.0, 1.0, 20.0, m)
        if it[0] == "f":
            return b.FlipNormal(make(it[1]))
        return b.Translate(b.Rotate(1, b.Cube((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), m), 10.0), (1.0, 1.0, 1.0))

    world = b.HittableList()
    for it in spec:
        world.push(make(it))
    b.set_scene(world, [])
    return b


def test_rooms_are_formed_from_exact_faces_only(pbe):
    """rt_flatten.cpp form_room: which rects of a list become a room, where it stands, what stays, and the tie-rule indices."""
    XY, XZ, YZ = Plane.XY, Plane.XZ, Plane.YZ
    rooms = lambda b: [o for o in R.debug_objects(b) if o["is_cube"] & 2]
    # three walls: not worth it; four: a room
    assert rooms(_room_list(pbe, [("w", XY, True), ("w", XZ, True), ("w", YZ, True)])) == []
    t = R.debug_objects(_room_list(pbe, [("w", XY, True), ("w", XZ, True), ("w", YZ, True), ("w", YZ, False)]))
    assert len(t) == 1 and t[0]["is_cube"] & 2 and t[0]["geom_count"] == 4 and t[0]["first_op"] == 0
    assert [(t[0]["is_cube"] >> (8 + 3 * f)) & 7 for f in range(6)] == [0, 7, 1, 7, 2, 3]
    # things between the walls are searched first, in their order; every wall learns the first of them that stood after it
    spec = [("c",), ("w", YZ, True), ("f", ("p", XY, True)), ("w", YZ, False), ("w", XZ, False), ("f", ("r",)), ("w", XY, True), ("c",)]
    t = R.debug_objects(_room_list(pbe, spec))
    assert [(o["n_ops"], bool(o["is_cube"] & 2)) for o in t] == [(2, False), (1, False), (1, False), (0, True), (2, False)]
    assert [(t[3]["first_op"] >> (5 * j)) & 31 for j in range(4)] == [1, 2, 2, 3]
    # ... but nothing with a Translate / Rotate may stand between the first and the last wall (world_hit asks the path's OWN ray whether a
    # plane distance could be NaN): no room then; before the first or after the last wall it may
    spec = [("c",), ("w", YZ, True), ("w", YZ, False), ("w", XZ, False), ("c",), ("w", XY, True)]
    assert rooms(_room_list(pbe, spec)) == []
    spec = [("c",), ("w", YZ, True), ("w", YZ, False), ("w", XZ, False), ("w", XY, True), ("c",)]
    assert len(rooms(_room_list(pbe, spec))) == 1
    # a patch in a wall's plane and a rect elsewhere are no faces: they stay, a run split around the walls that left it
    spec = [("w", YZ, True), ("p", YZ, True), ("w", YZ, False), ("r",), ("r",), ("w", XZ, False), ("w", XY, True)]
    t = R.debug_objects(_room_list(pbe, spec))
    assert [(o["geom_count"], bool(o["is_cube"] & 2)) for o in t] == [(1, False), (2, False), (4, True)]
    assert [(t[2]["first_op"] >> (5 * j)) & 31 for j in range(4)] == [0, 1, 2, 2]
    # a second wall on a face stays an ordinary rect (searched before the room: it stood before the last wall); a wrapped wall is no wall
    spec = [("w", YZ, True), ("w", YZ, True), ("w", YZ, False), ("f", ("w", XZ, True)), ("w", XZ, False), ("w", XY, True)]
    t = R.debug_objects(_room_list(pbe, spec))
    assert [(o["geom_count"], o["n_ops"], bool(o["is_cube"] & 2)) for o in t] == [(1, 0, False), (1, 1, False), (4, 0, True)]
    assert [(t[2]["first_op"] >> (5 * j)) & 31 for j in range(4)] == [0, 1, 2, 2]
    # a scene whose only other feature is a BVH of triangles (the mesh kernels) gets a room in the SIMPLE form only: the walls next to each
    # other, nothing between them — it stands where they stood, no second list; with the lamp between two walls the list stays as it is
    def mesh_room(lamp_between):
        bm = SceneBuilder(pbe)
        mm = bm.Lambertian(bm.ConstantTexture((0.5, 0.5, 0.5)))
        wl = bm.HittableList()
        walls = [bm.AARect(YZ, 0.0, 6.0, 0.0, 8.0, 4.0, mm), bm.AARect(YZ, 0.0, 6.0, 0.0, 8.0, 0.0, mm), bm.AARect(XZ, 0.0, 4.0, 0.0, 8.0, 0.0, mm),
                 bm.AARect(XZ, 0.0, 4.0, 0.0, 8.0, 6.0, mm), bm.AARect(XY, 0.0, 4.0, 0.0, 6.0, 8.0, mm)]
        lamp = bm.FlipNormal(bm.AARect(XZ, 1.0, 3.0, 1.0, 3.0, 5.9, bm.DiffuseLight(bm.ConstantTexture((4.0, 4.0, 4.0)))))
        for i, w_ in enumerate(walls):
            if lamp_between and i == 2:
                wl.push(lamp)
            wl.push(w_)
        if not lamp_between:
            wl.push(lamp)
        tris = bm.HittableList()
        for k in range(4):
            tris.push(bm.Triangle(((1.0, 1.0 + k, 2.0), (2.0, 1.0 + k, 2.0), (1.5, 1.5 + k, 3.0)), mm))
        wl.push(bm.BVH(tris, 0.0, 1.0))
        bm.set_scene(wl, [lamp])
        return bm
    t = R.debug_objects(mesh_room(False), top_only=False)
    assert [(o["geom_kind"], o["geom_count"], bool(o["is_cube"] & 2)) for o in t] == [(0, 5, True), (0, 1, False), (4, 1, False)] and t[0]["first_op"] == 0
    assert rooms(mesh_room(True)) == []
    # scenes the list-scene and mesh kernels do not serve keep their lists (a sphere: F_SPHERES)
    b = _room_list(pbe, [("w", XY, True), ("w", XZ, True), ("w", YZ, True), ("w", YZ, False)])
    b2 = SceneBuilder(pbe)
    m = b2.Lambertian(b2.ConstantTexture((0.5, 0.5, 0.5)))
    w = b2.HittableList()
    for pl, k in ((XY, 8.0), (XY, 0.0)):
        w.push(b2.AARect(pl, 0.0, 4.0, 0.0, 6.0, k, m))
    for pl, k in ((XZ, 6.0), (XZ, 0.0)):
        w.push(b2.AARect(pl, 0.0, 4.0, 0.0, 8.0, k, m))
    w.push(b2.Sphere((2.0, 2.0, 2.0), 1.0, m))
    b2.set_scene(w, [])
    assert rooms(b2) == [] and len(rooms(b)) == 1


def test_flatten_duplicated_handle_in_a_list(pbe):
    """`list.push(a.clone()); list.push(a)` is legal in the reference: the second occurrence must not stretch the first one's
    primitive range over a neighbouring (or the padding) record."""
    b = SceneBuilder(pbe)
    m = b.Lambertian(b.ConstantTexture((1, 1, 1)))
    a, c = b.Sphere((0, 0, 0), 1.0, m), b.Sphere((5, 0, 0), 1.0, m)
    inner = b.HittableList(); inner.push(a); inner.push(a)
    world = b.HittableList(); world.push(inner); world.push(c)
    b.set_scene(world, [])
    f = R.flatten(b)
    assert f["spheres"] == 2                  # a's record is shared
    assert f["objects"] == 3                  # [a], [a] again, [c]: no range covers a record it does not own
    # the same list without the duplicate merges into one run
    b2 = SceneBuilder(pbe)
    m2 = b2.Lambertian(b2.ConstantTexture((1, 1, 1)))
    w2 = b2.HittableList(); w2.push(b2.Sphere((0, 0, 0), 1.0, m2)); w2.push(b2.Sphere((5, 0, 0), 1.0, m2))
    b2.set_scene(w2, [])
    f2 = R.flatten(b2)
    assert (f2["spheres"], f2["objects"]) == (2, 1)


def test_sah_builder_keeps_the_table_shapes(pbe, earth):
    """The opt-in SAH builder only reshapes the tree: same leaves, same node count (one object per leaf), within the depth limit."""
    for name in ("random", "final", "teapot"):
        b = build_scene(name, pbe, earth)[0]
        ref = R.flatten(b)
        R.set_bvh_builder(b, R.RT_BVH_SAH)
        sah = R.flatten(b)
        assert sah == ref
        R.set_bvh_builder(b, R.RT_BVH_MEDIAN)
        assert R.flatten(b) == ref
    with pytest.raises(R.RenderError):
        R.set_bvh_builder(b, 7)


def test_launch_bookkeeping_before_any_launch(pbe):
    b = build_scene("cornell", pbe)[0]
    assert R.kernel_time_total(b) == (0.0, 0)                     # nothing launched, nothing to wait for
    for query in (R.last_stats, R.last_flush_count, R.last_traversal_stats, R.last_kernel_ms):
        with pytest.raises(R.RenderError):
            query(b)


def test_obj_loader_teapot():
    pos, idx = scenes.load_obj(scenes.asset_path("teapot.obj"), (0.0, 0.0, 0.0), 1.0)
    assert len(pos) == 530 and len(idx) == 3 * 1024 and max(idx) == 529 and min(idx) == 0
    import struct
    assert pos[0] == tuple(struct.unpack("f", struct.pack("f", x))[0] for x in (40.6266, 28.3457, -1.10804))   # f32 then widened
    assert idx[-3:] == [528, 529, 469]      # last face `f 529//529 530//530 470//470`


def test_error_paths_return_errors(pbe):
    b = SceneBuilder(pbe)
    m = b.Lambertian(b.ConstantTexture((1, 1, 1)))
    with pytest.raises(SceneError, match="no object in the scene"):      # src/bvh.rs:55 panics
        b.BVH([], 0.0, 1.0)
    assert pbe.lib.rt_sphere(b.h, (C.c_double * 3)(0, 0, 0), 1.0, 99) < 0
    assert b"bad material" in pbe.lib.rt_last_error()
    # a BVH child without a bounding box (bvh.rs:28,61 panic "no bounding box in bvh node"): an empty list, or a wrapper of one
    s = b.Sphere((0, 0, 0), 1.0, m)
    world = b.BVH([s, b.Translate(b.HittableList(), (1, 0, 0))], 0.0, 1.0)
    b.set_scene(world, [])
    with pytest.raises(R.RenderError, match="no bounding box in bvh node"):
        R.flatten(b)
    # BVHs inside BVH leaves: one level (rt_ir.h RT_MAX_NEST); deeper is refused, not mis-rendered
    b2 = SceneBuilder(pbe)
    m2 = b2.Lambertian(b2.ConstantTexture((1, 1, 1)))
    lvl2 = b2.BVH([b2.Sphere((0, 0, 0), 1.0, m2), b2.Sphere((3, 0, 0), 1.0, m2)], 0.0, 1.0)
    lvl1 = b2.BVH([lvl2, b2.Sphere((6, 0, 0), 1.0, m2)], 0.0, 1.0)
    b2.set_scene(b2.BVH([lvl1, b2.Sphere((9, 0, 0), 1.0, m2)], 0.0, 1.0), [])
    with pytest.raises(R.RenderError, match="RT_MAX_NEST"):
        R.flatten(b2)
    b2.set_scene(lvl1, [])
    assert R.flatten(b2)["bvh_nodes"] == 3 + 3
    # a ConstantMedium inside a ConstantMedium's boundary stays refused
    b4 = SceneBuilder(pbe)
    inner = b4.ConstantMedium(b4.Sphere((0, 0, 0), 1.0, b4.Dielectric(1.5)), 0.1, b4.ConstantTexture((1, 1, 1)))
    b4.set_scene(b4.ConstantMedium(b4.Translate(inner, (1, 0, 0)), 0.1, b4.ConstantTexture((1, 1, 1))), [])
    with pytest.raises(R.RenderError, match="nested ConstantMedium"):
        R.flatten(b4)
    b3 = SceneBuilder(pbe)
    with pytest.raises(R.RenderError, match="world not set"):
        R.flatten(b3)


def test_render_without_gpu_fails_loudly(pbe):
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    b, cam, bg = scenes.cornell_box(pbe)
    with pytest.raises(R.RenderError, match="no HIP device"):
        R.render(b, cam, bg, 8, 8, 1, 5)


def test_render_argument_checks(pbe):
    b, cam, bg = scenes.cornell_box(pbe)
    with pytest.raises(R.RenderError, match="W and H"):
        R.render(b, cam, bg, 1, 8, 1, 5)
    with pytest.raises(R.RenderError, match="samples_per_pixel"):
        R.render(b, cam, bg, 8, 8, 0, 5)


def test_local_tiles(pbe):
    assert R.local_tiles(800, 800, 64, 0, 1) == 10000
    assert R.local_tiles(800, 800, 64, 3, 8) == 1250
    assert R.local_tiles(10, 10, 64, 1, 8) == 1          # 2 real tiles, padded to 1 per rank
    assert R.local_tiles(1920, 1080, 64, 7, 8) == 4050


def test_jpeg_ingest_matches_an_independent_decoder():
    """csrc/rt_jpeg.cpp stands in for `image::open(..).to_rgb8()` (src/main.rs:248,491).  JPEG decoders legitimately differ
    by a few LSB (IDCT / colour-conversion rounding), so the check is against Pillow (libjpeg) with that tolerance."""
    from PIL import Image
    for name, max_ok in [("earthmap_256x128_444.jpg", 3), ("earthmap_256x128_grey.jpg", 2)]:
        path = golden_path(name)
        data, w, h = scenes.load_image_rgb8(path)
        ours = np.frombuffer(data, dtype=np.uint8).reshape(h, w, 3).astype(int)
        ref = np.asarray(Image.open(path).convert("RGB")).astype(int)
        assert ours.shape == ref.shape == (128, 256, 3)
        d = np.abs(ours - ref)
        assert d.max() <= max_ok and d.mean() < 0.05 and (d == 0).mean() > 0.99
    # and the decoded JPEG is the texture the PNG fixture holds, up to JPEG loss
    png = np.asarray(Image.open(golden_path("earthmap_256x128.png")).convert("RGB")).astype(int)
    assert np.abs(ours - png.mean(axis=2, keepdims=True)).mean() < 12      # grey JPEG vs the colour fixture's luma-ish mean


def test_jpeg_ingest_of_the_reference_asset_matches_an_independent_decoder():
    """The reference's earthmap.jpg itself (1024x512, baseline, 4:4:4; src/main.rs:491-495) through csrc/rt_jpeg.cpp vs Pillow."""
    from PIL import Image
    data, w, h = scenes.load_earthmap()
    ours = np.frombuffer(data, dtype=np.uint8).reshape(h, w, 3).astype(int)
    ref = np.asarray(Image.open(scenes.asset_path("earthmap.jpg")).convert("RGB")).astype(int)
    assert ours.shape == ref.shape == (512, 1024, 3)
    d = np.abs(ours - ref)
    assert d.max() <= 3 and d.mean() < 0.01 and (d.max(axis=2) == 0).mean() > 0.995


def test_jpeg_ingest_rejects_what_it_does_not_support(tmp_path):
    from PIL import Image
    im = Image.open(golden_path("earthmap_256x128.png"))
    sub = str(tmp_path / "sub420.jpg"); im.save(sub, quality=90, subsampling=2)
    prog = str(tmp_path / "prog.jpg"); im.save(prog, quality=90, subsampling=0, progressive=True)
    junk = str(tmp_path / "junk.jpg"); open(junk, "wb").write(b"not a jpeg at all")
    trunc = str(tmp_path / "trunc.jpg"); open(trunc, "wb").write(open(golden_path("earthmap_256x128_444.jpg"), "rb").read()[:300])
    for path, msg in [(sub, "subsampled"), (prog, "baseline"), (junk, "not a JPEG"), (trunc, "JPEG")]:
        with pytest.raises(RuntimeError, match=msg):
            scenes.load_image_rgb8(path)


def test_jpeg_restart_intervals(tmp_path):
    from PIL import Image
    im = Image.open(golden_path("earthmap_256x128.png"))
    p = str(tmp_path / "rst.jpg")
    try:
        im.save(p, quality=90, subsampling=0, restart_marker_blocks=7)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    if b"\xff\xdd" not in open(p, "rb").read():
        pytest.skip("this Pillow ignored restart_marker_blocks")
    data, w, h = scenes.load_image_rgb8(p)
    ours = np.frombuffer(data, dtype=np.uint8).reshape(h, w, 3).astype(int)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    assert np.abs(ours - ref).max() <= 3


def _bvh_with_children(be, kind):
    """A BVH of four bare primitives and three children of `kind` (tests of rt_bvh over every Hittable, bvh.rs:18-31); returns
    (builder, the BVH's handle)."""
    b = SceneBuilder(be)
    m = b.Lambertian(b.ConstantTexture((0.5, 0.5, 0.5)))
    glass = b.Dielectric(1.5)
    prims = [b.Sphere((0.0, 0.0, 0.0), 1.0, m), b.Cube((3.0, -1.0, -1.0), (5.0, 1.5, 1.0), m), b.Triangle([(7.0, 0.0, 0.0), (9.0, 0.5, 0.0), (8.0, 2.0, 1.0)], m),
             b.MovingSphere((-4.0, 0.0, 0.0), (-4.0, 0.5, 0.0), 0.0, 1.0, 0.8, m)]

    def one(i):
        base = b.Sphere((2.0 * i, 4.0, 1.0), 0.7, m) if i % 2 == 0 else b.Cube((2.0 * i, 3.0, 0.0), (2.0 * i + 1.0, 4.5, 1.5), m)
        if kind == "translate": return b.Translate(base, (0.5, -0.25, 2.0))
        if kind == "flip": return b.FlipNormal(base)
        if kind == "rotate": return b.Rotate(i % 3, base, 20.0 + 10.0 * i)
        if kind == "rotate_translate": return b.Translate(b.Rotate(1, base, 15.0), (1.0, 0.0, -1.0))
        if kind == "list":
            l = b.HittableList(); l.push(base); l.push(b.Translate(b.Sphere((2.0 * i, 6.0, 0.0), 0.5, m), (0.0, 0.0, 1.0))); l.push(b.AARect(0, 0.0, 1.0, 0.0, 1.0, 5.0 + i, m))
            return l
        if kind == "medium": return b.ConstantMedium(b.Translate(b.Sphere((2.0 * i, 4.0, 1.0), 0.9, glass), (0.0, 1.0, 0.0)), 0.5, b.ConstantTexture((1.0, 1.0, 1.0)))
        if kind == "bvh": return b.BVH([base, b.Sphere((2.0 * i + 0.5, 6.0, 0.0), 0.4, m), b.FlipNormal(b.Sphere((2.0 * i - 0.5, 7.0, 0.5), 0.3, m))], 0.0, 1.0)
        raise KeyError(kind)

    h = b.BVH(prims + [one(i) for i in range(3)], 0.0, 1.0)
    return b, h


@pytest.mark.parametrize("kind", ["translate", "flip", "rotate", "rotate_translate", "list", "medium", "bvh"])
def test_bvh_accepts_every_hittable_kind(kind, pbe, obe):
    """BVH::new takes Vec<Box<dyn Hittable>> and needs only bounding_box (bvh.rs:18-31,52-63): HittableList (hit.rs:73-88), FlipNormal
    (hit.rs:122-124), Translate (translate.rs:32-40), Rotate (rotate.rs:37-66,108-110: the whole-space box of quirk B3), ConstantMedium
    (medium.rs:63-65) and BVH itself (bvh.rs:93-95) all provide one.  Host side of it (the samples: tests/test_fuzz_gpu.py): such a child
    becomes a leaf of kind G_OBJ over the sub-objects it flattens to, the tree is the reference's (root box equal to the oracle's
    bounding_box of the same BVH, bit for bit), and the skip links thread all of it."""
    b, h = _bvh_with_children(pbe, kind)
    ob, oh = _bvh_with_children(obe, kind)
    world = b.HittableList(); world.push(b.Translate(h, (0.0, 0.0, 3.0))); world.push(b.AARect(1, -50.0, 50.0, -50.0, 50.0, -5.0, b.Lambertian(b.ConstantTexture((0.2, 0.2, 0.2)))))
    b.set_scene(world, [])
    c = R.flatten(b)
    n = c["bvh_nodes"]
    assert n == 2 * 7 - 1 + (3 * (2 * 3 - 1) if kind == "bvh" else 0)
    links = (C.c_uint32 * (4 * n))(); roots = (C.c_uint32 * 8)(); n_roots = C.c_uint32(0)
    assert pbe.lib.rt_debug_bvh_links(b.h, links, n, roots, 8, C.byref(n_roots)) == n
    L = np.frombuffer(links, np.uint32).reshape(n, 4)
    LEAF = 1 << 31
    leaf_kinds = ((L[:, 0] >> 28) & 7)[(L[:, 0] & LEAF) != 0]
    assert (leaf_kinds == 5).sum() == (3 + 3 if kind == "bvh" else 3)          # G_OBJ: the three children (and, nested, one FlipNormal(Sphere) in each inner BVH)
    assert n_roots.value == (4 if kind == "bvh" else 1)
    # every sub-object belongs to exactly one G_OBJ leaf: the leaves' runs partition [n_top, n_objects) (a nested BVH's own leaves put
    # their sub-objects into the table while the BVH is being built as a sub-object of the outer leaf: the runs must not interleave)
    objs, n_top = _objects(pbe, b)
    runs = sorted((int(a & 0x0FFFFFFF), int(cnt)) for a, cnt in L[((L[:, 0] & LEAF) != 0) & (((L[:, 0] >> 28) & 7) == 5)][:, :2])
    assert runs and runs[0][0] == n_top and all(x[0] + x[1] == y[0] for x, y in zip(runs, runs[1:])) and runs[-1][0] + runs[-1][1] == len(objs)
    assert sorted(r[1] for r in runs) == {"list": [3, 3, 3], "bvh": [1] * 6}.get(kind, [1, 1, 1])
    # 2 top-level objects; the sub-objects behind them: one per child (a list: its three items)
    assert c["objects"] == 2 + {"list": 9, "bvh": 3 + 3}.get(kind, 3)
    eb = np.zeros((n, 6)); fm = C.c_float(0)
    pbe.lib.rt_debug_filter_nodes.restype = C.c_int
    pbe.lib.rt_debug_filter_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    assert pbe.lib.rt_debug_filter_nodes(b.h, None, None, eb.ctypes.data, n, C.byref(fm)) == n
    box = np.zeros(6)
    assert obe.lib.orc_bounding_box(ob.h, oh.id, 0.0, 1.0, box.ctypes.data_as(C.POINTER(C.c_double))) == 1
    assert np.array_equal(eb[roots[0]].view(np.uint64), box.view(np.uint64)), (eb[roots[0]], box)
    if kind in ("rotate", "rotate_translate"):
        assert np.abs(box).max() > 1e308 and fm.value == 0.0                   # quirk B3: all of space; no f32 filter over such a tree (the exact walk)
    else:
        assert fm.value >= 1.0


@pytest.mark.parametrize("name", ["random", "final", "teapot"])
def test_bvh_skip_links_thread_the_recursions_order(name, pbe, earth):
    """The kernels walk a BVH in the reference's order without a stack: `node = hit && inner ? left : skip` (rt_kernel.hip bvh_hit_ww; the f64 kernels' filtered walk threads the same links, bvh_hit_filt).
    For ANY pattern of box-test outcomes that walk must meet the nodes BVH::hit's recursion meets (src/bvh.rs:77-91: bbox, left, right),
    in the same order.  Checked on the flattened trees of the shipped scenes with all-hit, all-miss and random outcome patterns."""
    be = pbe
    b, _, _ = build_scene(name, pbe, earth)
    n = R.flatten(b)["bvh_nodes"]
    links = (C.c_uint32 * (4 * n))(); roots = (C.c_uint32 * 8)(); n_roots = C.c_uint32(0)
    assert be.lib.rt_debug_bvh_links(b.h, links, n, roots, 8, C.byref(n_roots)) == n
    L = np.frombuffer(links, np.uint32).reshape(n, 4)
    LEAF, DONE = 1 << 31, 0xFFFFFFFF
    assert n_roots.value >= 1
    seen = np.zeros(n, bool)
    for root in list(roots)[:n_roots.value]:
        assert L[root, 3] == DONE
        for trial in range(4):
            rng = np.random.default_rng(trial)
            hit = {0: np.ones(n, bool), 1: np.zeros(n, bool)}.get(trial, rng.random(n) < (0.5 if trial == 2 else 0.9))
            want = []                                             # the recursion, iteratively: bbox test, then left, then right
            todo = [int(root)]
            while todo:
                i = todo.pop(); want.append(i)
                if hit[i] and not (L[i, 0] & LEAF): todo.append(int(L[i, 1])); todo.append(int(L[i, 2]))
            got, i = [], int(root)                                # the kernels' threaded walk
            while i != DONE:
                got.append(i)
                i = int(L[i, 2]) if (hit[i] and not (L[i, 0] & LEAF)) else int(L[i, 3])
                assert len(got) <= n
            assert got == want
            if trial == 0: seen[got] = True
    assert seen.all()                                             # every node belongs to exactly the trees walked


@pytest.mark.parametrize("name", ["random", "final", "teapot", "random tuned for its view"])
def test_filter_tree_is_a_conservative_hierarchy_over_the_same_leaves(name, pbe, earth):
    """The f64 kernels' box steps walk the FILTER tree (rt_ir.h DFNode; rt_flatten.cpp make_filter_nodes): f32 boxes and links from which
    near-duplicate inner nodes have been taken out.  What makes that exact (rt_kernel.hip: the ordered-scan form of BVH::hit) is checked
    here on the host: (1) every f32 box contains its node's f64 box (outward rounding) and filter_m bounds every coordinate; (2) with every
    box test passing the filter walk meets exactly the leaves of the reference tree, each once, in the reference's depth-first order;
    (3) every node the filter walk can stand at contains every leaf it reaches before that node's skip link — culling a node never
    skips a leaf outside it; (4) contraction did take nodes out (and never a leaf or a root)."""
    tuned = name.endswith("tuned for its view")
    b, cam, _ = build_scene(name.split()[0], pbe, earth)
    n = R.flatten(b)["bvh_nodes"]
    if tuned:
        # round 6: worlds that are ONE bare BVH get their contraction from a view's estimated pass rates (rt_flatten.cpp tune_filter_tree,
        # what rt_scene_calibrate and the synchronous renders do): the tree it leaves must pass the very same checks
        pbe.lib.rt_debug_tune_filter.restype = C.c_int
        pbe.lib.rt_debug_tune_filter.argtypes = [C.c_void_p, C.c_void_p]
        before = np.zeros((n, 2), np.uint32)
        pbe.lib.rt_debug_filter_nodes.restype = C.c_int
        pbe.lib.rt_debug_filter_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
        assert pbe.lib.rt_debug_filter_nodes(b.h, None, before.ctypes.data, None, n, None) == n
        assert pbe.lib.rt_debug_tune_filter(b.h, C.byref(cam)) == 1
        after = np.zeros((n, 2), np.uint32)
        assert pbe.lib.rt_debug_filter_nodes(b.h, None, after.ctypes.data, None, n, None) == n
        assert not np.array_equal(before, after)                               # another set of nodes left the tree
        for other in ("final", "teapot", "cornell"):                           # scenes whose BVHs stand beside other objects (or have none) are left alone
            ob, ocam, _ = build_scene(other, pbe, earth)
            R.flatten(ob)
            assert pbe.lib.rt_debug_tune_filter(ob.h, C.byref(ocam)) == 0
    links = (C.c_uint32 * (4 * n))(); roots = (C.c_uint32 * 8)(); n_roots = C.c_uint32(0)
    assert pbe.lib.rt_debug_bvh_links(b.h, links, n, roots, 8, C.byref(n_roots)) == n
    L = np.frombuffer(links, np.uint32).reshape(n, 4)
    fb = np.zeros((n, 6), np.float32); fl = np.zeros((n, 2), np.uint32); eb = np.zeros((n, 6)); fm = C.c_float(0)
    pbe.lib.rt_debug_filter_nodes.restype = C.c_int
    pbe.lib.rt_debug_filter_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    assert pbe.lib.rt_debug_filter_nodes(b.h, fb.ctypes.data, fl.ctypes.data, eb.ctypes.data, n, C.byref(fm)) == n
    LEAF, FLEAF, DONE = 1 << 31, 0x40000000, 0xFFFFFFFF
    # (1)
    assert (fb[:, 0::2].astype(np.float64) <= eb[:, :3]).all() and (fb[:, 1::2].astype(np.float64) >= eb[:, 3:]).all()
    assert fm.value >= max(1.0, float(np.abs(fb).max())) and fm.value <= 2.0 ** 40
    is_leaf = (L[:, 0] & LEAF) != 0
    assert np.array_equal((fl[:, 1] & FLEAF) != 0, is_leaf) and np.array_equal(fl[is_leaf, 1] & ~np.uint32(FLEAF), np.flatnonzero(is_leaf).astype(np.uint32))
    visited_inner = set()
    for root in list(roots)[:n_roots.value]:
        want = []                                                  # the reference's leaves in depth-first order (bvh.rs:77-91)
        todo = [int(root)]
        while todo:
            i = todo.pop()
            if L[i, 0] & LEAF: want.append(i)
            else: todo.append(int(L[i, 1])); todo.append(int(L[i, 2]))
        got, path, i = [], [], int(root)                           # (2) the filter walk, every test passing
        while i != DONE:
            path.append(i)
            if fl[i, 1] & FLEAF: got.append(i); i = int(fl[i, 0])
            else: visited_inner.add(i); i = int(fl[i, 1])
            assert len(path) <= 2 * n
        assert got == want and fl[root, 0] == DONE
        # (3) the leaves met between standing at node x and arriving at x's skip link all lie inside x's box
        pos = {x: k for k, x in enumerate(path)}
        for x in path:
            end = pos.get(int(fl[x, 0]), len(path)) if fl[x, 0] != DONE else len(path)
            below = [y for y in path[pos[x]:end] if fl[y, 1] & FLEAF]
            assert below, x
            bb = eb[below]
            assert (bb[:, :3].min(axis=0) >= fb[x, 0::2]).all() and (bb[:, 3:].max(axis=0) <= fb[x, 1::2]).all(), x
    # (4)
    n_inner = int((~is_leaf).sum())
    assert 0 < n_inner - len(visited_inner) < n_inner // 2, (n_inner, len(visited_inner))


def test_valu_op_weights_follow_from_the_committed_ubench_run():
    """The issue weights of the f64-VALU roofline (workloads.VALU_OP_WEIGHTS) are nothing but arithmetic on profiles/r04_ubench.csv, the
    tools/ubench output of the MI355X: a reader can recompute `roofline.frac` from files in the repository alone."""
    import csv
    from raytracinginrust_amd import workloads as W
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), W.UBENCH_CSV)
    rows = {r[0]: float(r[6]) for r in csv.reader(l for l in open(path) if not l.startswith("#")) if r[0] != "op"}
    assert W.weights_from_ubench(rows) == W.VALU_OP_WEIGHTS
    unit = 0.5 * (rows["add_f64"] + rows["mul_f64"])
    assert abs(unit - 4.92) < 0.01 and 4.0 < unit < 5.5          # the specification's 4 cycles per wave-instruction are not reached: both are stated
    ghz = {r[0]: float(r[5]) for r in csv.reader(l for l in open(path) if not l.startswith("#")) if r[0] != "op"}
    measured = 64.0 / unit * 0.5 * (ghz["add_f64"] + ghz["mul_f64"]) * 1e9 * 1024
    assert abs(measured - W.F64_VALU_MEASURED_ISSUE_OPS) / measured < 0.01
    assert W.F64_VALU_PEAK_OPS == 39.3e12
    for key, want in (("C1", 9846), ("C2", 2123), ("C3", 7731), ("C4", 4457), ("C5", 1402)):
        assert abs(W.valu_ops(W.F64_OPS_PER_SAMPLE[key]) - want) < 1.0
