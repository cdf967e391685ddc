"""GPU parity tests: the HIP path (through the C-ABI of librt_amd.so) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and — at BASELINE's full size — through
size-independent properties.

Tolerance (f64 kernel vs f64 oracle, same random stream): every +,-,*,/ and sqrt rounds identically on
both sides (-ffp-contract=off); the differences left are (a) sin/cos/atan2/acos/log of the device math
library vs glibc (<= a few ulp), (b) the kernel folds the throughput forward (beta *= w) while the
reference's recursion multiplies on the way back (main.rs:97), (c) the order of the per-pixel sum.  All are
O(depth * 1e-16) relative per sample, so:
    per sample : |gpu - oracle| <= 1e-9 * (1 + |oracle|)   (NaN/inf must match in kind)
    per pixel  : |gpu - oracle| <= 1e-9 * (spp + |oracle|)
A path whose branch decision flips on a last-ulp difference would break this for that one sample; none is
expected at these sizes (P ~ 1e-10 per sample) and at most MAX_DIVERGED are tolerated.
"""
import os

import numpy as np
import pytest

from conftest import build_scene, golden_path
from raytracinginrust_amd import _lib, dist as D, render as R, scenes
from raytracinginrust_amd.api import Axis, Camera, Plane, SceneBuilder

pytestmark = pytest.mark.gpu

SAMPLE_RTOL = 1e-9
MAX_DIVERGED = 2

CASES = {  # name: (W, H, spp, depth)   — sizes the oracle finishes in seconds
    "cornell": (48, 48, 32, 50),
    "random": (64, 36, 16, 8),        # BASELINE config 1 shape (16:9, depth 8)
    "final": (40, 40, 16, 50),
    "teapot": (64, 36, 16, 50),
}


def _compare_samples(gs, rs):
    """Returns (#diverged samples, max abs diff over the rest); NaN/inf patterns must be identical."""
    assert gs.shape == rs.shape
    assert np.array_equal(np.isnan(gs), np.isnan(rs))
    assert np.array_equal(np.isposinf(gs), np.isposinf(rs)) and np.array_equal(np.isneginf(gs), np.isneginf(rs))
    fin = np.isfinite(rs)
    d = np.where(fin, np.abs(np.where(fin, gs, 0.0) - np.where(fin, rs, 0.0)), 0.0)
    bad = (d > SAMPLE_RTOL * (1.0 + np.abs(np.where(fin, rs, 0.0)))).any(axis=-1)
    return int(bad.sum()), float(d[~bad].max()) if (~bad).any() else 0.0, bad


@pytest.fixture(scope="module")
def orc_mod():
    from oracle import orc
    orc.load()
    return orc


@pytest.mark.parametrize("name", list(CASES))
def test_scene_parity_per_sample_and_per_pixel(name, pbe, obe, orc_mod, earth):
    W, H, spp, depth = CASES[name]
    ob, ocam, obg = build_scene(name, obe, earth)
    pb, pcam, pbg = build_scene(name, pbe, earth)
    ref, rs, cnt = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True)
    n_bad, max_d, bad = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED, f"{n_bad} of {W * H * spp} samples diverged"
    clean = ~bad.any(axis=-1)                                   # pixels without a diverged sample
    fin = np.isfinite(ref)
    dp = np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0))
    assert np.all(dp[clean] <= SAMPLE_RTOL * (spp + np.abs(np.where(fin, ref, 0.0))[clean]))
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
    # the 8-bit image the reference would print (vec.rs:125-131): identical up to quantisation-boundary ties
    a, b = R.format_image(got, spp), R.format_image(ref, spp)
    assert (a != b).sum() <= 3 and np.abs(a.astype(int) - b.astype(int)).max() <= 1


@pytest.mark.parametrize("name", ["cornell", "random", "final", "teapot"])
def test_against_committed_golden(name, pbe, earth):
    g = np.load(golden_path(f"oracle_{name}.npz"))
    W, H, spp, depth, seed = int(g["W"]), int(g["H"]), int(g["spp"]), int(g["depth"]), int(g["seed"])
    pb, pcam, pbg = build_scene(name, pbe, earth)
    got = R.render(pb, pcam, pbg, W, H, spp, depth, seed=seed)
    ref = g["rgb_sum"]
    assert np.all(np.abs(got - ref) <= SAMPLE_RTOL * (spp + np.abs(ref)))


# ------------------------------------------------------------------ exact-value invariants on the GPU
def test_white_furnace_exact(pbe):
    b = SceneBuilder(pbe)
    rho = (0.25, 0.5, 0.75)
    world = b.HittableList()
    world.push(b.Sphere((0.0, 0.0, 0.0), 1.0, b.Lambertian(b.ConstantTexture(rho))))
    b.set_scene(world, [])
    cam = Camera((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 4.0, 0.0, 1.0)
    _, s = R.render(b, cam, (1.0, 1.0, 1.0), 64, 64, 16, 50, want_samples=True)
    s = s.reshape(-1, 3)
    hit = np.abs(s - np.array(rho)).max(axis=1) < 1e-14
    miss = (s == 1.0).all(axis=1)
    assert np.all(hit | miss) and hit.sum() > 1000 and miss.sum() > 1000


def test_glass_furnace_exact(pbe):
    b = SceneBuilder(pbe)
    world = b.HittableList()
    world.push(b.Sphere((0.0, 0.0, 0.0), 1.0, b.Dielectric(1.5)))
    b.set_scene(world, [])
    cam = Camera((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 4.0, 0.0, 1.0)
    _, s = R.render(b, cam, (1.0, 1.0, 1.0), 64, 64, 16, 50, want_samples=True)
    s = s.reshape(-1, 3)
    ones, zeros = (s == 1.0).all(axis=1), (s == 0.0).all(axis=1)
    assert np.all(ones | zeros) and ones.mean() > 0.999


# ------------------------------------------------------------------ edge cases
def _cornell(pbe):
    return scenes.cornell_box(pbe)


def test_minimum_frame_and_ragged_sample_counts(pbe, obe, orc_mod):
    """W = H = 2 (u,v divide by W-1, H-1), spp = 1 and spp not a multiple of the wave width."""
    for W, H, spp in [(2, 2, 1), (3, 2, 37), (5, 7, 65), (2, 9, 130)]:
        ob, ocam, obg = scenes.cornell_box(obe)
        pb, pcam, pbg = scenes.cornell_box(pbe)
        ref = orc_mod.render(ob, ocam, obg, W, H, spp, 50)
        got = R.render(pb, pcam, pbg, W, H, spp, 50)
        assert np.all(np.abs(got - ref) <= SAMPLE_RTOL * (spp + np.abs(ref))), (W, H, spp)


def test_depth_budget(pbe):
    b, cam, bg = _cornell(pbe)
    assert np.all(R.render(b, cam, bg, 16, 16, 8, 0) == 0.0)                    # main.rs:42-45
    _, s = R.render(b, cam, bg, 32, 32, 8, 1, want_samples=True)
    assert set(np.unique(s)) <= {0.0, 15.0}                                     # only the emitter is visible at depth 1


def test_nan_samples_match_reference_semantics(pbe, obe, orc_mod):
    """A `lights` entry with the trait-default pdf_value = 0 / random = (1,0,0) (hit.rs:29-30) makes the mixture
    pdf 0 for grazing directions: weight = att * 0 / 0 = NaN (Appendix B8).  The kernel must poison the same samples,
    and format_color must print them as 0."""
    def build(be):
        b = SceneBuilder(be)
        white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
        floor = b.AARect(Plane.XZ, -100.0, 100.0, -100.0, 100.0, 0.0, white)
        cube = b.Cube((-10.0, 0.0, -10.0), (10.0, 20.0, 10.0), white)       # Cube has no pdf_value/random of its own
        world = b.HittableList()
        world.push(floor)
        world.push(cube)
        b.set_scene(world, [cube])
        cam = Camera((0.0, 50.0, -120.0), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
        return b, cam, (0.5, 0.7, 1.0)
    ob, ocam, obg = build(obe)
    pb, pcam, pbg = build(pbe)
    ref, rs, cnt = orc_mod.render(ob, ocam, obg, 32, 32, 8, 10, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, 32, 32, 8, 10, want_samples=True)
    assert cnt["nonfinite"] > 100                                              # the case is actually exercised
    n_bad, _, _ = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED
    assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
    assert np.array_equal(R.format_image(got, 8), R.format_image(ref, 8))


def test_sphere_light_and_wrapped_instances(pbe, obe, orc_mod):
    """Sphere as a light (sphere.rs:104-119), Rotate about X and Z, FlipNormal outside a Translate, a Mesh list
    used directly (no BVH), a ConstantMedium around a rotated box."""
    def build(be):
        b = SceneBuilder(be)
        white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
        blue = b.Lambertian(b.ConstantTexture((0.2, 0.3, 0.8)))
        glow = b.DiffuseLight(b.ConstantTexture((9.0, 8.0, 7.0)))
        bulb = b.Sphere((0.0, 60.0, 0.0), 8.0, glow)
        world = b.HittableList()
        world.push(b.AARect(Plane.XZ, -100.0, 100.0, -100.0, 100.0, 0.0, white))
        world.push(bulb)
        world.push(b.Rotate(Axis.X, b.Cube((-30.0, 0.0, -10.0), (-10.0, 20.0, 10.0), blue), 20.0))
        world.push(b.Translate(b.Rotate(Axis.Z, b.Cube((0.0, 0.0, 0.0), (15.0, 25.0, 15.0), b.Metal((0.9, 0.9, 0.9), 0.2)), -25.0), (20.0, 0.0, -5.0)))
        world.push(b.FlipNormal(b.Translate(b.AARect(Plane.XY, -40.0, 40.0, 0.0, 50.0, 0.0, white), (0.0, 0.0, 60.0))))
        world.push(b.Mesh([(-50.0, 0.0, 30.0), (-30.0, 0.0, 30.0), (-40.0, 30.0, 35.0), (-40.0, 10.0, 10.0)], [0, 1, 2, 0, 2, 3, 1, 2, 3], blue))
        smoke_box = b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (20.0, 20.0, 20.0), white), 30.0), (-5.0, 0.0, -40.0))
        world.push(b.ConstantMedium(smoke_box, 0.05, b.ConstantTexture((1.0, 1.0, 1.0))))
        b.set_scene(world, [bulb])
        cam = Camera((0.0, 70.0, -160.0), (0.0, 20.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.5, 170.0, 0.0, 1.0)
        return b, cam, (0.02, 0.02, 0.05)
    ob, ocam, obg = build(obe)
    pb, pcam, pbg = build(pbe)
    ref, rs = orc_mod.render(ob, ocam, obg, 48, 48, 16, 30, want_samples=True)
    got, gs = R.render(pb, pcam, pbg, 48, 48, 16, 30, want_samples=True)
    n_bad, _, _ = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED
    assert ref.mean() > 0.0


@pytest.mark.parametrize("with_lights", [True, False])
def test_principled_material_parity(pbe, obe, orc_mod, with_lights):
    """PBR + PDF::BRDF + the Microfacet arm (mat.rs:84-197, pdf.rs:97-160, main.rs:99-105), incl. its NaN samples."""
    def build(be):
        b = SceneBuilder(be)
        a = b.PBR(b.ConstantTexture((0.8, 0.3, 0.2)), 0.2, 0.1, 0.5, 0.4, 0.3, 0.2, 0.3, 0.5, 0.6, 0.8)
        c = b.PBR(b.ConstantTexture((0.9, 0.9, 0.9)), 1.0, 0.0, 0.5, 0.15, 0.0, 0.6, 0.0, 0.0, 0.0, 0.0)
        d = b.PBR(b.CheckTexture(b.ConstantTexture((0.2, 0.8, 0.3)), b.ConstantTexture((0.9, 0.9, 0.2))), 0.0, 0.8, 0.2, 0.9, 0.5, 0.0, 1.0, 0.5, 1.0, 0.2)
        light = b.DiffuseLight(b.ConstantTexture((10.0, 10.0, 10.0)))
        rect_light = b.FlipNormal(b.AARect(Plane.XZ, -20.0, 20.0, -20.0, 20.0, 60.0, light))
        world = b.HittableList()
        world.push(b.Sphere((-22.0, 10.0, 0.0), 10.0, a))
        world.push(b.Sphere((0.0, 10.0, 5.0), 10.0, c))
        world.push(b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (14.0, 18.0, 14.0), d), 25.0), (14.0, 0.0, -8.0)))
        world.push(b.AARect(Plane.XZ, -100.0, 100.0, -100.0, 100.0, 0.0, b.Lambertian(b.ConstantTexture((0.7, 0.7, 0.7)))))
        world.push(rect_light)
        b.set_scene(world, [rect_light] if with_lights else [])
        cam = Camera((0.0, 35.0, -90.0), (0.0, 10.0, 0.0), (0.0, 1.0, 0.0), 35.0, 1.0, 0.5, 95.0, 0.0, 1.0)
        return b, cam, (0.1, 0.1, 0.15)
    ob, ocam, obg = build(obe)
    pb, pcam, pbg = build(pbe)
    ref, rs, cnt = orc_mod.render(ob, ocam, obg, 40, 40, 16, 12, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, 40, 40, 16, 12, want_samples=True)
    n_bad, _, _ = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED
    assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
    assert np.isfinite(rs).all(axis=-1).mean() > 0.5


def _other_scene(name, be, earth):
    if name == "two_spehre":
        return scenes.two_spehre(be)
    if name == "two_perlin_sphere":
        return scenes.two_perlin_sphere(be)
    if name == "earth":
        return scenes.earth(be, *earth)
    if name == "light_room":
        return scenes.light_room(be)
    if name == "cornell_box_with_smoke":
        return scenes.cornell_box_with_smoke(be)
    if name == "progress_showcase":
        return scenes.progress_showcase(be)
    raise KeyError(name)


@pytest.mark.parametrize("name,scatter", [("two_spehre", False), ("two_perlin_sphere", False), ("earth", False), ("light_room", False),
                                          ("cornell_box_with_smoke", False), ("cornell_box_with_smoke", True), ("progress_showcase", False)])
def test_remaining_reference_scenes(pbe, obe, orc_mod, earth, name, scatter):
    """The six scene functions no BASELINE config names (src/main.rs:212-276,313-346,515-562), and the opt-in
    RT_ISOTROPIC_SCATTER mode on the smoke scene."""
    ob, ocam, obg = _other_scene(name, obe, earth)
    pb, pcam, pbg = _other_scene(name, pbe, earth)
    W, H, spp, depth = 48, 27, 8, 20
    if scatter:
        orc_mod.set_isotropic_scatters(ob, True)
    ref, rs = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True, flags=R.RT_ISOTROPIC_SCATTER if scatter else R.RT_F64)
    n_bad, _, _ = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED
    if name == "progress_showcase":
        assert np.all(got == 0.0)                      # empty world, black background
    if name == "cornell_box_with_smoke" and scatter:
        absorb = R.render(pb, pcam, pbg, W, H, spp, depth)
        assert got.mean() > absorb.mean()              # scattering media return light that absorbing media swallow


@pytest.mark.parametrize("name", ["random", "final", "teapot"])
def test_near_first_bvh_mode_gives_the_same_samples(name, pbe, obe, orc_mod, earth):
    """RT_NEAR_FIRST_BVH (opt-in) visits the nearer child first; the closest hit — and therefore every sample — is the
    reference's, ties included (resolved by preorder).  Checked against the oracle, not just against the default mode."""
    W, H, spp, depth = CASES[name]
    ob, ocam, obg = build_scene(name, obe, earth)
    pb, pcam, pbg = build_scene(name, pbe, earth)
    ref, rs = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True)
    got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True, flags=R.RT_NEAR_FIRST_BVH)
    n_bad, _, _ = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED


def test_prepare_does_not_change_the_render(pbe):
    from raytracinginrust_amd import scenes
    b1, cam, bg = scenes.cornell_box(pbe)
    b2, _, _ = scenes.cornell_box(pbe)
    R.prepare(b2)
    R.prepare(b2, R.RT_F32)
    assert np.array_equal(R.render(b1, cam, bg, 24, 24, 4, 8, seed=3), R.render(b2, cam, bg, 24, 24, 4, 8, seed=3))
    with pytest.raises(R.RenderError):
        R.prepare(SceneBuilder(pbe))                     # no world set


def _mesh_room(be, seed):
    """A lit room (rect list) with two triangle-mesh BVHs (one inside Translate(Rotate(..))) and a loose triangle: only
    list + BVH + triangle features, i.e. the mesh kernel."""
    rs = np.random.RandomState(seed)
    b = SceneBuilder(be)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    red = b.Lambertian(b.ConstantTexture((0.65, 0.05, 0.05)))
    steel = b.Metal((0.8, 0.85, 0.88), 0.1)
    light = b.DiffuseLight(b.ConstantTexture((12.0, 12.0, 12.0)))
    world = b.HittableList()
    lamp = b.FlipNormal(b.AARect(Plane.XZ, 150.0, 400.0, 150.0, 400.0, 554.0, light))
    world.push(lamp)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, red))

    def blob(center, radius, n, mat):
        tris = []
        for _ in range(n):
            p0 = center + rs.uniform(-radius, radius, 3)
            tris.append(b.Triangle([tuple(p0), tuple(p0 + rs.uniform(-25, 25, 3)), tuple(p0 + rs.uniform(-25, 25, 3))], mat))
        return b.BVH(tris, 0.0, 1.0)

    world.push(blob(np.array([200.0, 120.0, 250.0]), 70.0, int(rs.randint(40, 160)), white))
    world.push(b.Triangle([(20.0, 20.0, 300.0), (120.0, 30.0, 280.0), (60.0, 200.0, 350.0)], steel))
    world.push(b.Translate(b.Rotate(Axis.Y, blob(np.array([0.0, 0.0, 0.0]), 60.0, int(rs.randint(30, 120)), steel), float(rs.uniform(-40, 40))),
                           (380.0, 200.0, 300.0)))
    b.set_scene(world, [lamp])
    cam = Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.02, 0.02, 0.03)


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("extra", [0, R.RT_NEAR_FIRST_BVH])
def test_persistent_traversal_is_scheduling_only(pbe, obe, orc_mod, seed, extra):
    """RT_PERSISTENT_BVH / RT_LOCKSTEP_BVH change which lanes run when, never what a path computes: every sample is
    bit-identical between the two loop shapes, and both match the oracle."""
    b, cam, bg = _mesh_room(pbe, seed)
    W, H, spp, depth = 40, 40, 8, 16
    _, lock = R.render(b, cam, bg, W, H, spp, depth, seed=5 + seed, flags=R.RT_LOCKSTEP_BVH | extra, want_samples=True)
    assert R.last_traversal_stats(b)["traversal_steps"] == 0
    _, pers = R.render(b, cam, bg, W, H, spp, depth, seed=5 + seed, flags=R.RT_PERSISTENT_BVH | extra, want_samples=True)
    tv = R.last_traversal_stats(b)
    assert tv["traversal_steps"] > 0 and 0 < tv["traversal_lanes"] <= 64 * tv["traversal_steps"]
    assert 0 < tv["leaf_steps"] < tv["traversal_steps"] and 0 < tv["leaf_lanes"] <= 64 * tv["leaf_steps"]
    assert np.array_equal(lock.view(np.uint64), pers.view(np.uint64))
    _, auto = R.render(b, cam, bg, W, H, spp, depth, seed=5 + seed, flags=extra, want_samples=True)
    # two small mesh BVHs beside a list: the lock-step loop by default (the persistent loop is chosen for trees of some size only — 640
    # nodes x the square of the number of BVH objects, measured in round 4: tools/mesh_size_probe.py; the teapot room's 2047-node tree gets it)
    assert R.last_traversal_stats(b)["traversal_steps"] == 0
    assert np.array_equal(auto.view(np.uint64), pers.view(np.uint64))
    if extra == 0:
        ob, ocam, obg = _mesh_room(obe, seed)
        _, ref = orc_mod.render(ob, ocam, obg, W, H, spp, depth, seed=5 + seed, want_samples=True)
        assert np.array_equal(np.isnan(pers), np.isnan(ref))
        fin = np.isfinite(ref)
        d = np.abs(np.where(fin, pers, 0.0) - np.where(fin, ref, 0.0))
        assert (d > 1e-9 * (1.0 + np.abs(np.where(fin, ref, 0.0)))).any(axis=-1).sum() <= 1


def test_persistent_traversal_on_the_teapot_room(pbe):
    b, cam, bg = scenes.cornell_test(pbe, scenes.asset_path("teapot.obj"))
    W, H, spp, depth = 64, 64, 4, 50
    _, lock = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    _, pers = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    assert R.last_traversal_stats(b)["traversal_steps"] > 0
    assert np.array_equal(lock.view(np.uint64), pers.view(np.uint64))
    # the schedule is a tuning knob: any setting gives the same samples
    R.set_traversal_schedule(b, 64, 8, 40)
    _, odd = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    assert np.array_equal(odd.view(np.uint64), pers.view(np.uint64))
    with pytest.raises(R.RenderError):
        R.set_traversal_schedule(b, 16, 32, 16)
    # a BVH that is the whole world stays on the lock-step loop unless asked otherwise
    mesh_only = SceneBuilder(pbe)
    lam = mesh_only.Lambertian(mesh_only.ConstantTexture((0.5, 0.5, 0.5)))
    tris = [mesh_only.Triangle([(float(i), 0.0, 0.0), (float(i) + 1.0, 0.0, 0.0), (float(i), 1.0, 0.0)], lam) for i in range(8)]
    mesh_only.set_scene(mesh_only.BVH(tris, 0.0, 1.0), [])
    cam2 = Camera((4.0, 0.5, -10.0), (4.0, 0.5, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    _, a = R.render(mesh_only, cam2, (0.5, 0.7, 1.0), 16, 16, 4, 5, want_samples=True)
    assert R.last_traversal_stats(mesh_only)["traversal_steps"] == 0
    _, c = R.render(mesh_only, cam2, (0.5, 0.7, 1.0), 16, 16, 4, 5, flags=R.RT_PERSISTENT_BVH, want_samples=True)
    assert R.last_traversal_stats(mesh_only)["traversal_steps"] > 0
    assert np.array_equal(a.view(np.uint64), c.view(np.uint64))


@pytest.mark.parametrize("name", ["random", "final", "teapot"])
def test_sah_builder_finds_the_same_hits(name, pbe, earth):
    """RT_BVH_SAH changes the shape of the tree, not what a ray hits: per-sample agreement with the reference-shaped tree
    (a sample may differ only through an exactly-equal-t tie or a last-ulp box cull, as with RT_NEAR_FIRST_BVH)."""
    b, cam, bg = build_scene(name, pbe, earth)
    W, H, spp, depth = 48, 48, 8, 12
    _, ref = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    R.set_bvh_builder(b, R.RT_BVH_SAH)
    for flags in (0, R.RT_NEAR_FIRST_BVH):
        _, got = R.render(b, cam, bg, W, H, spp, depth, flags=flags, want_samples=True)
        same = (got.view(np.uint64) == ref.view(np.uint64)).all(axis=-1)
        assert same.mean() > 0.999, f"{name}: {int((~same).sum())} of {same.size} samples differ"


def test_stop_on_zero_flag_is_equivalent_without_nans(pbe):
    b, cam, bg = _cornell(pbe)
    a = R.render(b, cam, bg, 64, 64, 32, 50)
    c = R.render(b, cam, bg, 64, 64, 32, 50, flags=R.RT_STOP_ON_ZERO)
    assert np.all(np.abs(a - c) <= 1e-12 * (32 + np.abs(a)))


# ------------------------------------------------------------------ tile sharding on one GPU
@pytest.mark.parametrize("world,tile_px", [(1, 64), (3, 64), (8, 100), (2, 7)])
def test_tile_sharded_render_reassembles_to_full_frame(pbe, world, tile_px):
    import torch
    b, cam, bg = _cornell(pbe)
    W, H, spp, depth = 50, 30, 16, 50
    full = R.render(b, cam, bg, W, H, spp, depth)
    parts = []
    for rank in range(world):
        tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, tile_px=tile_px, rank=rank, world=world)
        parts.append(tr.render_local().clone())
    torch.cuda.synchronize()
    frame = D.assemble(torch.stack(parts, 0), W, H, tile_px).cpu().numpy()
    assert np.all(np.abs(frame - full) <= 1e-12 * (spp + np.abs(full)))


# ------------------------------------------------------------------ BASELINE's full size: size-independent properties
def test_two_frames_in_flight_give_the_same_frames(pbe):
    """TileRenderer(pipeline=2): consecutive frames alternate between two streams / buffers / launch slots of one scene."""
    import torch
    b, cam, bg = _cornell(pbe)
    W, H, spp, depth = 96, 80, 32, 20
    one = D.TileRenderer(b, cam, bg, W, H, spp, depth, rank=0, world=1, pipeline=1)
    ref = one.render_frame().clone(); one.sync()
    two = D.TileRenderer(b, cam, bg, W, H, spp, depth, rank=0, world=1, pipeline=2)
    R.kernel_time_total(b, reset=True)
    frames = []
    for i in range(5):
        f = two.render_frame()                         # valid once its own stream is done: copy it on that stream
        with torch.cuda.stream(two.streams[i % 2]):
            frames.append(f.clone())
    two.sync()
    ms, n = R.kernel_time_total(b)
    assert n == 5 and ms > 0.0
    for f in frames:
        assert torch.allclose(f, ref, rtol=1e-12, atol=1e-12 * spp, equal_nan=True)


_FULL = {}


def _full_frame(key, pbe, earth):
    """The BASELINE config's whole frame at its full sample count through ONE launch (what bench.py times), rendered once per test
    session: (frame sums, kernel ms, non-finite samples, builder, camera, background)."""
    if key not in _FULL:
        from raytracinginrust_amd import workloads
        w = workloads.WORKLOADS[key]
        b, cam, bg = workloads.build(w, pbe, earth)
        a = R.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth)
        _FULL[key] = (a, R.last_kernel_ms(b), R.last_stats(b)["nonfinite_samples"], b, cam, bg)
    return _FULL[key]


@pytest.mark.parametrize("band", [0, 1], ids=["middle", "top"])
@pytest.mark.parametrize("key", ["C2", "C3", "C4", "C5"])
def test_full_spp_band_against_the_oracle(key, band, pbe, earth):
    """BASELINE configs 2-5, the WHOLE frame at the FULL sample count through one launch (the bench's step), compared per pixel with the
    oracle on two bands of rows — 8 / 2 / 2 / 1 rows in the middle of the image and the top row — whose oracle sums at full spp are
    committed (tests/golden/oracle_band_*.npz, make_golden.py: make_band_goldens; the CPU suite checks that the oracle still produces
    them).  This is what ties samples 16..1023 / 4..4095 / 2..2047 / 1..8191 of a pixel, the 256-sample tail chunks (32 per pixel at
    C5) and the pixel x spp > 2^32 bookkeeping of the full-frame launch to the oracle: tolerance per pixel 1e-9 * (spp + |ref|), at
    most MAX_DIVERGED pixels may hold a diverged sample (a path that took another branch after a last-ulp libm difference), non-finite
    pixels equal."""
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS[key]
    g = np.load(golden_path(f"oracle_band_{key}.npz"))
    assert (int(g["W"]), int(g["H"]), int(g["spp"]), int(g["depth"])) == (w.W, w.H, w.spp, w.max_depth)
    a = _full_frame(key, pbe, earth)[0]
    r0, r1 = (int(x) for x in g["rows"][band])
    ref, got = g[f"band{band}"], a[r0:r1]
    assert ref.shape == got.shape == (r1 - r0, w.W, 3)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    d = np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0))
    bad = (d > SAMPLE_RTOL * (w.spp + np.abs(np.where(fin, ref, 0.0)))).any(axis=-1)
    assert bad.sum() <= MAX_DIVERGED, f"{int(bad.sum())} of {bad.size} pixels of rows {r0}..{r1 - 1} differ from the oracle at full spp (worst {d.max():.3e})"
    print(f"{key} rows {r0}..{r1 - 1} at {w.spp} spp: {int(bad.sum())} pixels off, max |gpu - oracle| per pixel sum {d[~bad].max():.3e} "
          f"(sums up to {np.abs(ref[fin]).max():.1f})")


def test_full_size_cornell_properties(pbe, obe, orc_mod):
    """BASELINE config 2: Cornell box 800x800, 1024 spp, depth 50 (the oracle link at full spp: test_full_spp_band_against_the_oracle).
    (1) two runs agree to summation-order rounding (the dynamic sample->lane assignment changes only the order);
    (2) no non-finite sample; (3) a sharded render (8 ranks) reassembles to the same frame."""
    import torch
    W = H = 800
    spp, depth = 1024, 50
    a, ms, n_bad, b, cam, bg = _full_frame("C2", pbe, None)
    assert n_bad == 0
    c = R.render(b, cam, bg, W, H, spp, depth)
    assert np.all(np.abs(a - c) <= 1e-12 * (spp + np.abs(a)))
    parts = []
    for rank in range(8):
        tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, tile_px=64, rank=rank, world=8)
        parts.append(tr.render_local().clone())
    torch.cuda.synchronize()
    frame = D.assemble(torch.stack(parts, 0), W, H, 64).cpu().numpy()
    assert np.all(np.abs(frame - a) <= 1e-12 * (spp + np.abs(a)))
    print(f"C2 800x800x1024 f64: {ms:.1f} ms, {W * H * spp / ms / 1e3:.0f} Msamples/s")


def test_full_size_teapot_properties(pbe, obe, orc_mod):
    """BASELINE config 4 on one GPU: teapot room 1920x1080, 2048 spp, depth 50 (4.25 G samples).  (1) the persistent-traversal
    loop (default here) and the lock-step loop give the same frame to summation-order rounding; (2) no non-finite sample;
    (3) 8 interleaved shards reassemble to the frame.  (The oracle link at full spp: test_full_spp_band_against_the_oracle.)"""
    import torch
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS["C4"]
    a, ms, n_bad, b, cam, bg = _full_frame("C4", pbe, None)
    assert n_bad == 0
    c = R.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth, flags=R.RT_LOCKSTEP_BVH)
    assert np.all(np.abs(a - c) <= 1e-12 * (w.spp + np.abs(a)))
    del c
    parts = []
    for rank in range(8):
        tr = D.TileRenderer(b, cam, bg, w.W, w.H, w.spp, w.max_depth, tile_px=D.DEFAULT_TILE_PX, rank=rank, world=8)
        parts.append(tr.render_local().clone())
    torch.cuda.synchronize()
    frame = D.assemble(torch.stack(parts, 0), w.W, w.H, D.DEFAULT_TILE_PX).cpu().numpy()
    assert np.all(np.abs(frame - a) <= 1e-12 * (w.spp + np.abs(a)))
    print(f"C4 {w.W}x{w.H}x{w.spp} f64: {ms:.0f} ms, {w.samples / ms / 1e3:.0f} Msamples/s")


def test_full_size_final_scene_properties(pbe, obe, orc_mod, earth):
    """BASELINE config 3: final scene 800x800, 4096 spp, depth 50 (2.62 G samples).  Two runs agree to summation-order rounding
    with the same pixels poisoned by non-finite samples (the reference's 0/0 cases).  (The oracle link at full spp:
    test_full_spp_band_against_the_oracle.)"""
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS["C3"]
    a, ms, n_bad, b, cam, bg = _full_frame("C3", pbe, earth)
    c = R.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth)
    assert R.last_stats(b)["nonfinite_samples"] == n_bad
    assert np.array_equal(np.isfinite(a), np.isfinite(c))
    fin = np.isfinite(a)
    assert (~fin).any(axis=-1).sum() <= n_bad                    # every poisoned pixel holds at least one counted sample
    assert np.all(np.abs(a[fin] - c[fin]) <= 1e-12 * (w.spp + np.abs(a[fin])))
    print(f"C3 {w.W}x{w.H}x{w.spp} f64: {ms:.0f} ms, {w.samples / ms / 1e3:.0f} Msamples/s, {n_bad} non-finite samples")


def test_c1_full_size_per_sample_against_the_oracle(pbe, obe, orc_mod):
    """BASELINE config 1 at its full size — random_scene 400x225, 64 spp, depth 8 (5.76 M samples; the reference's CPU-runnable
    case, src/main.rs:153-210 with the 16:9 frame BASELINE names): every sample against the oracle, same tolerance as the small cases."""
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS["C1"]
    pb, pcam, pbg = workloads.build(w, pbe)
    ob, ocam, obg = workloads.build(w, obe)
    ref, rs, cnt = orc_mod.render(ob, ocam, obg, w.W, w.H, w.spp, w.max_depth, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, w.W, w.H, w.spp, w.max_depth, want_samples=True)
    n_bad, max_d, bad = _compare_samples(gs, rs)
    assert n_bad <= MAX_DIVERGED, f"{n_bad} of {w.samples} samples diverged"
    assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
    clean = ~bad.any(axis=-1)
    fin = np.isfinite(ref)
    dp = np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0))
    assert np.all(dp[clean] <= SAMPLE_RTOL * (w.spp + np.abs(np.where(fin, ref, 0.0))[clean]))
    a, c = R.format_image(got, w.spp), R.format_image(ref, w.spp)
    assert (a != c).sum() <= 5 and np.abs(a.astype(int) - c.astype(int)).max() <= 1
    print(f"C1 {w.W}x{w.H}x{w.spp}: {n_bad} diverged, max |gpu - oracle| per sample {max_d:.2e}, "
          f"{(gs.view(np.uint64) == rs.view(np.uint64)).all(axis=-1).mean():.3f} of samples bit-identical")


GRID_SPP = {"C2": 16, "C3": 4, "C4": 2, "C5": 1}     # samples 0 .. k-1 of every pixel of the config's own frame (tests/golden/make_golden.py uses the same)


def _block_sums(img, blk=16):
    """(H, W, 3) -> sums over blk x blk pixel blocks (ragged edge blocks included) of the finite values, and the non-finite count per block."""
    H, W, _ = img.shape
    fin = np.isfinite(img).all(axis=-1)
    v = np.where(fin[..., None], img, 0.0)
    ys, xs = np.arange(0, H, blk), np.arange(0, W, blk)
    sums = np.add.reduceat(np.add.reduceat(v, ys, axis=0), xs, axis=1)
    bad = np.add.reduceat(np.add.reduceat((~fin).astype(np.int64), ys, axis=0), xs, axis=1)
    return sums, bad


@pytest.mark.parametrize("key", ["C2", "C3", "C4", "C5"])
def test_config_grid_per_sample_against_the_oracle(key, pbe, obe, orc_mod, earth):
    """BASELINE configs 2-5 on their OWN pixel grids (800x800, 800x800, 1920x1080, 3840x2160; aspect, camera and pixel -> sample mapping of
    the real frame), at the first k samples of every pixel — the same RNG streams as samples 0..k-1 of the full config (rt_rng.h keys a
    path by (pixel, sample)), i.e. the BASELINE frame itself at reduced spp: EVERY sample against the oracle with the small cases'
    tolerance, then the per-pixel sums, the non-finite count, and the committed block sums of the oracle's frame (tests/golden/)."""
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS[key]
    spp = GRID_SPP[key]
    pb, pcam, pbg = workloads.build(w, pbe, earth)
    ob, ocam, obg = workloads.build(w, obe, earth)
    ref, rs, cnt = orc_mod.render(ob, ocam, obg, w.W, w.H, spp, w.max_depth, want_samples=True, want_counters=True)
    got, gs = R.render(pb, pcam, pbg, w.W, w.H, spp, w.max_depth, want_samples=True)
    n_bad, max_d, n_same, bad_px = 0, 0.0, 0, np.zeros((w.H, w.W), dtype=bool)
    for r0 in range(0, w.H, 128):                                   # in bands of rows: bounded temporaries
        nb, md, bad = _compare_samples(gs[r0:r0 + 128], rs[r0:r0 + 128])
        n_bad += nb; max_d = max(max_d, md); bad_px[r0:r0 + 128] = bad.any(axis=-1)
        n_same += int((gs[r0:r0 + 128].view(np.uint64) == rs[r0:r0 + 128].view(np.uint64)).all(axis=-1).sum())
    assert n_bad <= MAX_DIVERGED, f"{n_bad} of {w.W * w.H * spp} samples diverged"
    assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    dp = np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0))
    assert np.all(dp[~bad_px] <= SAMPLE_RTOL * (spp + np.abs(np.where(fin, ref, 0.0))[~bad_px]))
    g = np.load(golden_path(f"oracle_grid_{key}.npz"))
    assert (int(g["W"]), int(g["H"]), int(g["spp"]), int(g["depth"])) == (w.W, w.H, spp, w.max_depth)
    sums, nonfin = _block_sums(got)
    assert np.array_equal(nonfin, g["nonfinite"])
    if n_bad == 0:
        assert np.all(np.abs(sums - g["block_sums"]) <= SAMPLE_RTOL * (256 * spp + np.abs(g["block_sums"])))
    print(f"{key} {w.W}x{w.H} at samples 0..{spp - 1}: {n_bad} of {w.W * w.H * spp} diverged, max |gpu - oracle| per sample {max_d:.2e}, "
          f"{n_same / (w.W * w.H * spp):.3f} of samples bit-identical, {cnt['nonfinite']} non-finite")


def test_c5_full_sample_count(pbe, obe, orc_mod):
    """BASELINE config 5 at its real sample count: Cornell box 3840x2160 (src/main.rs:579-583 scaled), 8192 spp, depth 50 =
    67.9 G samples (pixels x spp > 2^32; 32 work chunks per pixel).  The whole frame on ONE GPU, then rank 0's share of the
    8-GPU decomposition at the same spp: (1) no non-finite sample; (2) the share's tiles equal the whole frame's pixels to
    summation-order rounding (sharded == unsharded at full spp); (3) the share is reproducible.  (The oracle link at full spp:
    test_full_spp_band_against_the_oracle.)"""
    import torch
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS["C5"]
    assert w.W * w.H * w.spp > 2 ** 32
    full, ms, n_bad, b, cam, bg = _full_frame("C5", pbe, None)
    assert n_bad == 0 and np.isfinite(full).all()
    tile = D.DEFAULT_TILE_PX
    tr = D.TileRenderer(b, cam, bg, w.W, w.H, w.spp, w.max_depth, tile_px=tile, rank=0, world=8)
    share = tr.render_local().clone(); torch.cuda.synchronize()
    ms_share = R.last_kernel_ms(b)
    assert R.last_stats(b)["nonfinite_samples"] == 0
    again = tr.render_local().clone(); torch.cuda.synchronize()
    share, again = share.cpu().numpy(), again.cpu().numpy()
    assert np.all(np.abs(share - again) <= 1e-12 * (w.spp + np.abs(share)))
    flat = full.reshape(-1, 3)
    n_px = w.W * w.H
    for q, t in enumerate(D.local_tile_ids(w.W, w.H, tile, 0, 8)):
        lo, hi = t * tile, min(n_px, (t + 1) * tile)
        if lo >= n_px:
            assert not share[q].any()                            # padding tile
            continue
        assert np.all(np.abs(share[q, : hi - lo] - flat[lo:hi]) <= 1e-12 * (w.spp + np.abs(flat[lo:hi])))
    print(f"C5 {w.W}x{w.H}x{w.spp} f64 on one GPU: {ms / 1e3:.2f} s, {w.samples / ms / 1e3:.0f} Msamples/s; rank 0's 1/8 share {ms_share:.0f} ms "
          f"(x8 = {8 * ms_share / 1e3:.2f} s)")


def test_f32_variant_statistical_parity(pbe):
    """RT_F32 is the throughput variant: same estimator in f32, so only statistical agreement is claimed."""
    b, cam, bg = _cornell(pbe)
    a = R.render(b, cam, bg, 200, 200, 256, 50) / 256
    c = R.render(b, cam, bg, 200, 200, 256, 50, flags=R.RT_F32) / 256
    assert np.all(np.isfinite(c))
    assert c.mean() == pytest.approx(a.mean(), rel=0.03)
    blocks = lambda x: x.reshape(10, 20, 10, 20, 3).mean(axis=(1, 3))
    assert np.abs(blocks(a) - blocks(c)).max() < 0.06


# ------------------------------------------------------------------ the C++ host (`main.rs` restated) end to end
@pytest.mark.parametrize("scene,depth", [("cornell", 20), ("random", 8), ("two_perlin", 8), ("smoke", 20)])
def test_cxx_host_main_prints_the_same_ppm(tmp_path, pbe, scene, depth):
    """host/rtrender builds the scene with the C++ mirror of the Rust API (its own random_scene draws included) and
    prints the reference's P3 stream; it must equal the Python-built scene's image up to quantisation ties."""
    import subprocess
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "host", "rtrender")
    assert os.path.exists(exe), "host/rtrender missing: run make -C raytracinginrust_amd/csrc"
    W, H, spp = 48, 27, 8
    txt = subprocess.run([exe, "--scene", scene, "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", str(depth)],
                         check=True, capture_output=True, text=True).stdout.split("\n")
    assert txt[:3] == ["P3", f"{W} {H}", "255"]
    got = np.array([[int(x) for x in l.split()] for l in txt[3:3 + W * H]]).reshape(H, W, 3)
    build = {"cornell": scenes.cornell_box, "random": scenes.random_scene, "two_perlin": scenes.two_perlin_sphere, "smoke": scenes.cornell_box_with_smoke}[scene]
    pb, pcam, pbg = build(pbe, aspect_ratio=W / H)
    ref = R.format_image(R.render(pb, pcam, pbg, W, H, spp, depth), spp)
    assert (got != ref).sum() <= 3 and np.abs(got - ref.astype(int)).max() <= 1


def test_cxx_host_final_scene_with_jpeg_ingest(pbe):
    """`rtrender --scene final --earth <baseline JPEG>`: the C++ host decodes the texture itself (image::open, main.rs:491),
    draws the scene's random numbers in the reference's order (boxes, Perlin::new, spheres) and must print the image the
    Python-built scene gives with the same decoded texels."""
    import subprocess
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "host", "rtrender")
    jpg = scenes.asset_path("earthmap.jpg")                  # the reference's own 1024x512 asset (src/main.rs:491-495)
    W, H, spp, depth = 40, 40, 8, 20
    txt = subprocess.run([exe, "--scene", "final", "--earth", jpg, "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", str(depth)],
                         check=True, capture_output=True, text=True).stdout.split("\n")
    got = np.array([[int(x) for x in l.split()] for l in txt[3:3 + W * H]]).reshape(H, W, 3)
    pb, pcam, pbg = scenes.final_scene(pbe, *scenes.load_image_rgb8(jpg))
    ref = R.format_image(R.render(pb, pcam, pbg, W, H, spp, depth), spp)
    assert (got != ref).sum() <= 3 and np.abs(got - ref.astype(int)).max() <= 1


def test_4k_frame_index_ranges(pbe):
    """BASELINE config 5's frame (3840x2160 = 8.3 M pixels; pixels x spp exceeds 2^32 at its 8192 spp): the 64-bit work
    arithmetic at that frame size with a small spp.  Sharded as on 8 GPUs (prime tile size, see dist.py) the assembled
    frame must equal the unsharded render, and the per-rank shares must be balanced."""
    import torch
    b, cam, bg = scenes.cornell_box(pbe, aspect_ratio=3840 / 2160)
    W, H, spp, depth = 3840, 2160, 4, 50
    full = R.render(b, cam, bg, W, H, spp, depth)
    assert R.last_stats(b)["nonfinite_samples"] == 0
    parts, means = [], []
    for rank in range(8):
        tr = D.TileRenderer(b, cam, bg, W, H, spp, depth, rank=rank, world=8)
        parts.append(tr.render_local().clone())
        means.append(float(parts[-1].mean().item()))
    torch.cuda.synchronize()
    frame = D.assemble(torch.stack(parts, 0), W, H, D.DEFAULT_TILE_PX).cpu().numpy()
    assert np.all(np.abs(frame - full) <= 1e-12 * (spp + np.abs(full)))
    assert max(means) / min(means) < 1.03                  # no aliasing between tile columns and ranks
    small = R.render(b, cam, bg, W // 8, H // 8, 64, depth)
    assert full.mean() / spp == pytest.approx(small.mean() / 64, rel=0.05)


@pytest.mark.parametrize("name", ["random", "final", "teapot"])
def test_lds_node_cache_is_scheduling_only(name, pbe, earth, monkeypatch):
    """The BVH kernels stage the top levels of the trees in LDS (one workgroup per CU); where a node is fetched from changes no
    sample.  RT_NODE_CACHE_MAX limits the staged nodes (0 = none, 100 = a partial cache)."""
    b, cam, bg = build_scene(name, pbe, earth)
    W, H, spp, depth = 64, 48, 8, 30
    _, full = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    info = R.last_launch_info(b)
    assert info["bvh_nodes_in_lds"] > 0 and info["workgroups_per_cu"] == 1 and info["threads"] in (768, 1024)
    for limit in ("0", "100"):
        monkeypatch.setenv("RT_NODE_CACHE_MAX", limit)
        _, part = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
        assert R.last_launch_info(b)["bvh_nodes_in_lds"] == min(int(limit), info["bvh_nodes"])
        assert np.array_equal(full.view(np.uint64), part.view(np.uint64))
    monkeypatch.delenv("RT_NODE_CACHE_MAX")


def test_aabb_hit_on_the_device_against_the_oracle(pbe, obe):
    """AABB::hit (src/aabb.rs:19-36) on the device vs the oracle's, on random and on hostile inputs: rays through box corners and
    along faces, zero direction components (1/d = inf, 0 * inf = NaN which f64::max / min ignore), infinite / NaN origins, inverted
    boxes.  The traversal's NaN-free form must agree with the reference's form wherever it may be used (tame ray, min <= max)."""
    import ctypes as C
    rnd = np.random.default_rng(7)
    n = 20000
    lo = rnd.uniform(-10, 10, (n, 3)); ext = rnd.uniform(0, 5, (n, 3))
    boxes = np.concatenate([lo, lo + ext], axis=1)
    o = rnd.uniform(-20, 20, (n, 3)); d = rnd.normal(size=(n, 3))
    tl = np.stack([np.full(n, 1e-5), rnd.choice([np.inf, 5.0, 50.0], n)], axis=1)
    special = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-320, 1e300, -1e300, 1e308]
    for i in range(0, n, 4):                 # every 4th case gets hostile components
        k = rnd.integers(0, 3)
        which = rnd.integers(0, 6)
        if which == 0: d[i, k] = rnd.choice([0.0, -0.0])
        elif which == 1: o[i, k] = boxes[i, k]; d[i, (k + 1) % 3] = 0.0          # origin on the min face, axis-parallel: 0 * inf
        elif which == 2: o[i, k] = rnd.choice(special)
        elif which == 3: d[i, k] = rnd.choice(special)
        elif which == 4: boxes[i, k], boxes[i, 3 + k] = boxes[i, 3 + k], boxes[i, k] - 1.0   # inverted box
        else: o[i] = boxes[i, :3]; d[i] = boxes[i, 3:] - boxes[i, :3]             # through two corners
    rays = np.concatenate([o, d], axis=1)
    out = np.zeros(n, dtype=np.int32)
    lib = pbe.lib
    lib.rt_debug_aabb_hit.restype = C.c_int
    lib.rt_debug_aabb_hit.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    boxes, rays, tl = (np.ascontiguousarray(x, dtype=np.float64) for x in (boxes, rays, tl))
    assert lib.rt_debug_aabb_hit(n, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, out.ctypes.data) == 0
    from oracle import orc
    d3 = lambda v: orc._d(*v)
    ref = np.array([obe.lib.orc_aabb_hit(d3(boxes[i, :3]), d3(boxes[i, 3:]), d3(rays[i, :3]), d3(rays[i, 3:]), tl[i, 0], tl[i, 1]) for i in range(n)])
    assert np.array_equal(out & 1, ref), f"{int(((out & 1) != ref).sum())} of {n} box tests differ from the oracle"
    box_tame = (boxes[:, :3] <= boxes[:, 3:]).all(axis=1) & np.isfinite(boxes).all(axis=1)
    use = ((out & 4) != 0) & box_tame
    assert use.sum() > n // 2 and (~use).sum() > 100
    assert np.array_equal((out[use] >> 1) & 1, ref[use]), "the NaN-free form disagrees where it would be used"
    # the filtered walk's f32 box step (rt_kernel.hip: filter_pass) is CONSERVATIVE: wherever it would be used it lets through
    # every box the exact test passes (it may let through more: the exact test runs on the leaf's own box afterwards)
    filt = use & ((out & 8) != 0)
    assert filt.sum() > n // 2
    assert not ((ref[filt] == 1) & ((out[filt] & 16) == 0)).any(), "the f32 filter culled a box AABB::hit passes"
    plain = filt & (np.arange(n) % 4 != 0)                     # on ordinary rays it culls nearly everything the exact test culls
    extra = ((ref[plain] == 0) & ((out[plain] & 16) != 0)).sum()
    assert extra <= 0.001 * plain.sum(), f"the f32 filter passes {extra} boxes of {plain.sum()} that AABB::hit culls"


def _cloud_room(be, n_tris, radius):
    """A lit room with ONE sparse cloud of small triangles in a corner (tools/mesh_size_probe.py's family)."""
    rs = np.random.RandomState(n_tris)
    b = SceneBuilder(be)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    light = b.DiffuseLight(b.ConstantTexture((12.0, 12.0, 12.0)))
    world = b.HittableList()
    lamp = b.FlipNormal(b.AARect(Plane.XZ, 150.0, 400.0, 150.0, 400.0, 554.0, light))
    world.push(lamp)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    tris = []
    for _ in range(n_tris):
        p0 = np.array([200.0, 120.0, 250.0]) + rs.uniform(-radius, radius, 3)
        tris.append(b.Triangle([tuple(p0), tuple(p0 + rs.uniform(-8, 8, 3)), tuple(p0 + rs.uniform(-8, 8, 3))], white))
    world.push(b.BVH(tris, 0.0, 1.0))
    b.set_scene(world, [lamp])
    return b, Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0), (0.02, 0.02, 0.03)


def test_loop_shape_of_a_mesh_scene_is_measured(pbe, monkeypatch):
    """Mesh scenes run a persistent-traversal or a lock-step loop — same samples, and which is faster depends on the view, not on the
    tree's size (round 5: the teapot room's 2047-node tree prefers the persistent loop by 10 %, a 1999-node cloud of triangles the
    lock-step loop by 10 %).  A calibration (rt_host.cpp: calibrate_loop_shape) measures both on a smaller copy of the view and keeps the
    faster FOR THAT VIEW: rt_scene_calibrate, or the first frame of >= 1e8 samples through a synchronous entry point; the asynchronous
    entry points never calibrate; small frames, other views and RT_NO_LOOP_CALIBRATION keep the size rule; rt_scene_set_loop_shape
    overrides everything; rt_last_loop_info says what ran and why.  (Functional assertions only: which shape is faster on the day is
    tools/mesh_size_probe.py's business, not a test's.)"""
    W, H, spp, depth = 1024, 1024, 256, 50
    for make in (lambda: scenes.cornell_test(pbe, scenes.asset_path("teapot.obj")), lambda: _cloud_room(pbe, 1000, 70.0)):
        b2, cam2, bg2 = make()                                     # a fresh scene: nothing measured yet
        _, auto = R.render(b2, cam2, bg2, 96, 96, 4, depth, want_samples=True)            # a small frame first: no calibration, the size rule (persistent: >= 640 nodes)
        li = R.last_loop_info(b2)
        assert li["shape"] == "persistent" and li["chosen_by"] == "size rule" and li["calibration_ms"] is None and li["feats"] == 261
        assert li["kernel"] == "rt::pathtrace_kernel<double, 261u>" and R.last_traversal_stats(b2)["traversal_steps"] > 0
        for _ in range(3):
            R.render(b2, cam2, bg2, W, H, spp, depth)              # the first of them measures the view (synchronous entry point, 2.7e8 samples)
        li = R.last_loop_info(b2)
        ms = li["calibration_ms"]
        assert li["chosen_by"] == "calibration of this view" and ms["lock-step"] > 0 and ms["persistent"] > 0
        if abs(ms["persistent"] - ms["lock-step"]) > 0.03 * min(ms.values()):             # the stored numbers decide, reproducibly
            assert (li["shape"] == "persistent") == (ms["persistent"] < ms["lock-step"]), li
        assert (li["shape"] == "persistent") == (R.last_traversal_stats(b2)["traversal_steps"] > 0) and R.stored_loop_shape(b2) in (0, 1)
        total_ms, n = R.kernel_time_total(b2)
        assert n == 4                                               # the calibration launches are not in the caller's totals
        # the calibration belongs to the view it measured: another frame size is the size rule's again
        _, lock = R.render(b2, cam2, bg2, 96, 96, 4, depth, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
        assert R.last_loop_info(b2)["chosen_by"] == "caller's flag" and R.last_loop_info(b2)["shape"] == "lock-step" and R.last_loop_info(b2)["feats"] == 5
        assert np.array_equal(auto.view(np.uint64), lock.view(np.uint64))
        R.render(b2, cam2, bg2, 96, 96, 4, depth)
        assert R.last_loop_info(b2)["chosen_by"] == "size rule"
        # a set shape serves every view; -1 forgets it (and the calibration)
        R.set_loop_shape(b2, 0)
        _, forced = R.render(b2, cam2, bg2, 96, 96, 4, depth, want_samples=True)
        li = R.last_loop_info(b2)
        assert li["shape"] == "lock-step" and li["chosen_by"] == "rt_scene_set_loop_shape" and np.array_equal(auto.view(np.uint64), forced.view(np.uint64))
        R.set_loop_shape(b2, -1)
        assert R.stored_loop_shape(b2) == -1
        monkeypatch.setenv("RT_NO_LOOP_CALIBRATION", "1")
        b3, cam3, bg3 = make()
        R.render(b3, cam3, bg3, W, H, spp, depth)
        assert R.last_traversal_stats(b3)["traversal_steps"] > 0    # the size rule: 1999 / 2047 nodes >= 640
        assert R.last_loop_info(b3)["chosen_by"] == "size rule"
        monkeypatch.delenv("RT_NO_LOOP_CALIBRATION")
        # the asynchronous entry points never measure: a fresh scene's large frame through rt_render_multi_device is the size rule's ...
        b4, cam4, bg4 = make()
        R.render_multi_device(b4, cam4, bg4, W, H, spp, depth, device_mask=1); R.multi_sync(b4)
        assert R.last_loop_info(b4)["chosen_by"] == "size rule" and R.kernel_time_total(b4)[1] == 1
        # ... until the caller calibrates the view (explicitly, synchronously): then they run what it found
        R.calibrate(b4, cam4, bg4, W, H, spp, depth)
        assert R.kernel_time_total(b4)[1] == 1                      # (its four launches stay out of the totals)
        R.render_multi_device(b4, cam4, bg4, W, H, spp, depth, device_mask=1); R.multi_sync(b4)
        li4 = R.last_loop_info(b4)
        assert li4["chosen_by"] == "calibration of this view" and li4["calibration_ms"]["persistent"] > 0
    # a scene whose loop shape is not open: nothing to measure, nothing stored
    b5, cam5, bg5 = scenes.cornell_box(pbe)
    R.calibrate(b5, cam5, bg5, W, H, spp, depth)
    R.render(b5, cam5, bg5, 64, 64, 4, depth)
    li5 = R.last_loop_info(b5)
    assert li5 == {"shape": "list", "feats": 0, "kernel": "rt::pathtrace_kernel<double, 0u>", "chosen_by": "the scene leaves no choice", "calibration_ms": None}


def test_view_tuned_filter_tree_is_scheduling_only(pbe, monkeypatch):
    """Worlds that are ONE bare BVH (random spheres) get the contraction of their filter tree from a view's estimated pass rates — at the
    first synchronous render of a view, or by rt_scene_calibrate (rt_host.cpp tune_for_view, rt_flatten.cpp tune_filter_tree).  Any
    conservative hierarchy over the leaves gives the reference's samples: every sample bit-identical with the area rule's tree
    (RT_NO_FILTER_TUNING), for the view the tree was tuned for and for another one; the asynchronous entry point keeps the tree it finds."""
    import ctypes as C
    W, H, spp, depth = 64, 48, 8, 8
    lib = pbe.lib
    lib.rt_debug_filter_nodes.restype = C.c_int
    lib.rt_debug_filter_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]

    def links(b):
        n = R.flatten(b)["bvh_nodes"]
        fl = np.zeros((n, 2), np.uint32)
        assert lib.rt_debug_filter_nodes(b.h, None, fl.ctypes.data, None, n, None) == n
        return fl

    monkeypatch.setenv("RT_NO_FILTER_TUNING", "1")
    b0, cam0, bg0 = scenes.random_scene(pbe, aspect_ratio=W / H)
    _, plain = R.render(b0, cam0, bg0, W, H, spp, depth, want_samples=True)
    area_tree = links(b0)
    monkeypatch.delenv("RT_NO_FILTER_TUNING")
    b1, cam1, bg1 = scenes.random_scene(pbe, aspect_ratio=W / H)
    assert np.array_equal(links(b1), area_tree)                                # as flattened: the area rule
    _, tuned = R.render(b1, cam1, bg1, W, H, spp, depth, want_samples=True)   # the first render of a view tunes the tree for it
    assert not np.array_equal(links(b1), area_tree)
    assert np.array_equal(plain.view(np.uint64), tuned.view(np.uint64))
    assert R.last_loop_info(b1)["feats"] == 2111
    # another view: re-tuned (synchronous entry point), same samples as the area rule's tree gives for that view
    cam2 = Camera((3.0, 6.0, 13.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 35.0, W / H, 0.0, 10.0, 0.0, 1.0)
    t1 = links(b1)
    _, tuned2 = R.render(b1, cam2, bg1, W, H, spp, depth, want_samples=True)
    assert not np.array_equal(links(b1), t1)
    monkeypatch.setenv("RT_NO_FILTER_TUNING", "1")                             # (read at every tune)
    _, plain2 = R.render(b0, cam2, bg0, W, H, spp, depth, want_samples=True)
    assert np.array_equal(links(b0), area_tree)
    monkeypatch.delenv("RT_NO_FILTER_TUNING")
    assert np.array_equal(plain2.view(np.uint64), tuned2.view(np.uint64))
    # the asynchronous entry point never tunes: a fresh scene rendered through it keeps the area rule's tree until the caller calibrates
    b3, cam3, bg3 = scenes.random_scene(pbe, aspect_ratio=W / H)
    R.render_multi_device(b3, cam3, bg3, W, H, spp, depth, device_mask=1); R.multi_sync(b3)
    assert np.array_equal(links(b3), area_tree)
    R.calibrate(b3, cam3, bg3, W, H, spp, depth)
    assert not np.array_equal(links(b3), area_tree)
    got = R.render_multi(b3, cam3, bg3, W, H, spp, depth, device_mask=1)
    assert np.all(np.abs(got - plain.sum(axis=2)) <= 1e-12 * (spp + np.abs(plain.sum(axis=2))))


def _walled_mesh_room(be, seed):
    """A room whose walls ARE faces of one box and stand next to each other in the list (what rt_flatten.cpp's form_room makes ONE object of
    in a mesh scene: the teapot room's shape, main.rs:416-444), 4 – 5 of them, the lamp behind them, then triangle-mesh BVHs (one under
    Translate(Rotate)), sometimes a bare Cube: the mesh kernels' single site of the Cube fast path."""
    rs = np.random.RandomState(300 + seed)
    b = SceneBuilder(be)
    mats = [b.Lambertian(b.ConstantTexture(tuple(float(x) for x in rs.uniform(0.1, 0.9, 3)))) for _ in range(3)] + [b.Metal((0.8, 0.85, 0.88), 0.1)]
    light = b.DiffuseLight(b.ConstantTexture((9.0, 9.0, 8.0)))
    L = 555.0
    faces = [(Plane.YZ, L), (Plane.YZ, 0.0), (Plane.XZ, 0.0), (Plane.XZ, L), (Plane.XY, L), (Plane.XY, 0.0)]
    keep = [faces[i] for i in rs.permutation(6)[: int(rs.choice([4, 5, 5]))]]
    world = b.HittableList()
    for plane, k_ in keep:
        world.push(b.AARect(plane, 0.0, L, 0.0, L, k_, mats[rs.randint(0, 4)]))
    lamp = b.FlipNormal(b.AARect(Plane.XZ, 128.0, 428.0, 115.0, 270.0, 554.0, light))
    world.push(lamp)

    def blob(center, radius, n, mat):
        tris = []
        for _ in range(n):
            p0 = center + rs.uniform(-radius, radius, 3)
            tris.append(b.Triangle([tuple(p0), tuple(p0 + rs.uniform(-30, 30, 3)), tuple(p0 + rs.uniform(-30, 30, 3))], mat))
        return b.BVH(tris, 0.0, 1.0)

    world.push(blob(np.array([200.0, 140.0, 260.0]), 80.0, int(rs.randint(60, 200)), mats[0]))
    if rs.rand() < 0.5:
        world.push(b.Cube((330.0, 0.0, 120.0), (430.0, 160.0, 220.0), mats[1]))
    world.push(b.Translate(b.Rotate(Axis.Y, blob(np.array([0.0, 0.0, 0.0]), 60.0, int(rs.randint(40, 120)), mats[3]), float(rs.uniform(-40, 40))), (380.0, 260.0, 330.0)))
    b.set_scene(world, [lamp])
    frm = [(278.0, 278.0, -800.0), (199.0, 439.0, -200.0), (278.0, 300.0, 40.0), (900.0, 700.0, -600.0)][seed % 4]
    cam = Camera(frm, (278.0, 250.0, 278.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.02, 0.02, 0.03)


@pytest.mark.parametrize("seed", list(range(8)))
def test_room_in_a_mesh_scene(pbe, obe, orc_mod, seed, monkeypatch):
    """The SIMPLE room form (round 6): walls that are faces of one box and stand next to each other in the list of a scene whose only other
    feature is a BVH of triangles — the teapot room, C4 — are one object where they stood, tested through the Cube fast path at the mesh
    kernels' world-list site (rt_kernel.hip RoomSite).  Per sample against the oracle, in both loop shapes, and word for word against the
    same scene flattened without the room (RT_NO_ROOM)."""
    mk = (lambda be: build_scene("teapot", be)) if seed == 0 else (lambda be: _walled_mesh_room(be, seed))
    W, H, spp, depth = 64, 48, 8, 30
    pb, pcam, pbg = mk(pbe)
    assert any(o["is_cube"] & 2 for o in R.debug_objects(pb)), "no room formed"
    ob, ocam, obg = mk(obe)
    _, rs_, cnt = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True, want_counters=True)
    shots = {}
    for name, flags in (("persistent", R.RT_PERSISTENT_BVH), ("lockstep", R.RT_LOCKSTEP_BVH)):
        _, gs = R.render(pb, pcam, pbg, W, H, spp, depth, flags=flags, want_samples=True)
        assert R.last_loop_info(pb)["feats"] in (5, 261)
        n_bad, _, _ = _compare_samples(gs, rs_)
        assert n_bad <= MAX_DIVERGED, f"{name}: {n_bad} of {W * H * spp} samples diverged"
        assert R.last_stats(pb)["nonfinite_samples"] == cnt["nonfinite"]
        shots[name] = gs
    assert np.array_equal(shots["persistent"].view(np.uint64), shots["lockstep"].view(np.uint64))
    monkeypatch.setenv("RT_NO_ROOM", "1")
    qb, qcam, qbg = mk(pbe)
    assert not any(o["is_cube"] & 2 for o in R.debug_objects(qb))
    _, plain = R.render(qb, qcam, qbg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    assert np.array_equal(plain.view(np.uint64), shots["lockstep"].view(np.uint64)), "the room form changed a sample"


def _big_mesh_room(be, n_tris):
    rs = np.random.RandomState(5)
    b = SceneBuilder(be)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    light = b.DiffuseLight(b.ConstantTexture((12.0, 12.0, 12.0)))
    world = b.HittableList()
    lamp = b.FlipNormal(b.AARect(Plane.XZ, 150.0, 400.0, 150.0, 400.0, 554.0, light))
    world.push(lamp)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    tris = []
    for _ in range(n_tris):
        p0 = np.array([278.0, 200.0, 250.0]) + rs.uniform(-150, 150, 3)
        tris.append(b.Triangle([tuple(p0), tuple(p0 + rs.uniform(-12, 12, 3)), tuple(p0 + rs.uniform(-12, 12, 3))], white))
    world.push(b.BVH(tris, 0.0, 1.0))
    b.set_scene(world, [lamp])
    cam = Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.02, 0.02, 0.03)


def test_tree_larger_than_the_lds(pbe, obe, orc_mod, monkeypatch):
    """A tree of 9999 nodes does not fit the CU's LDS even as 32-byte filter nodes: the top levels are staged, the rest comes from
    memory, node states are ids instead of LDS addresses (rt_kernel.hip: fetch_fnode<false>).  Both loop shapes, the cache on / off:
    every sample bit-identical, and equal to the oracle's."""
    b, cam, bg = _big_mesh_room(pbe, 5000)
    W, H, spp, depth = 48, 36, 4, 12
    _, pers = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_PERSISTENT_BVH, want_samples=True)
    info = R.last_launch_info(b)
    assert info["bvh_nodes"] == 9999 and 0 < info["bvh_nodes_in_lds"] < info["bvh_nodes"]
    _, lock = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH, want_samples=True)
    assert np.array_equal(pers.view(np.uint64), lock.view(np.uint64))
    monkeypatch.setenv("RT_NODE_CACHE_MAX", "0")
    _, none = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    monkeypatch.delenv("RT_NODE_CACHE_MAX")
    assert np.array_equal(pers.view(np.uint64), none.view(np.uint64))
    ob, ocam, obg = _big_mesh_room(obe, 5000)
    _, ref = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True)
    n_div, _, _ = _compare_samples(pers, ref)
    assert n_div <= MAX_DIVERGED


def test_box_filter_is_conservative_on_grazing_rays(pbe):
    """The f32 box step of the filtered walk must never cull a box whose exact AABB::hit passes (rt_kernel.hip: the proof above
    make_filter).  Aimed at where a one-sided error would show: rays through points ON faces, edges and corners moved by 1e-15 ... 1e-3
    of the box size, boxes far from the origin (coordinates up to 1e6: the f32 grid is coarse there), thin boxes (the reference's
    0.0001-thick rect boxes), and [t_min, t_max] ending within a few ulps of the entry / exit distance.  Checked against the exact
    form on the device, which test_aabb_hit_on_the_device_against_the_oracle ties to the oracle."""
    import ctypes as C
    rnd = np.random.default_rng(11)
    n = 400000
    scale = 10.0 ** rnd.uniform(-2, 6, (n, 1))
    lo = rnd.uniform(-1, 1, (n, 3)) * scale
    ext = rnd.uniform(0, 1, (n, 3)) * scale * 10.0 ** rnd.uniform(-4, 0, (n, 1))
    thin = rnd.integers(0, 4, n) == 0
    ext[thin, rnd.integers(0, 3, thin.sum())] = 0.0002
    boxes = np.concatenate([lo, lo + ext], axis=1)
    # a point on the box surface: each coordinate at min, max or inside; moved by a tiny relative offset
    w = rnd.choice([0.0, 1.0, 0.5, 0.25], (n, 3), p=[0.35, 0.35, 0.2, 0.1])
    tgt = lo + w * ext
    tgt += rnd.choice([-1.0, 0.0, 1.0], (n, 3)) * 10.0 ** rnd.uniform(-15, -3, (n, 3)) * (np.abs(tgt) + ext)
    o = tgt + rnd.normal(size=(n, 3)) * scale * 10.0 ** rnd.uniform(-3, 1, (n, 1))
    d = tgt - o
    d *= 10.0 ** rnd.uniform(-3, 3, (n, 1))                                   # |d| is not 1 in the reference either
    t_hit = np.linalg.norm(tgt - o, axis=1) / np.maximum(np.linalg.norm(d, axis=1), 1e-300)
    ulps = rnd.choice([-4, -1, 0, 1, 4, 1000], n)
    tmax = np.where(rnd.integers(0, 2, n) == 0, np.inf, t_hit * (1.0 + ulps * 2.0 ** -52))
    tmin = np.where(rnd.integers(0, 3, n) == 0, t_hit * (1.0 + rnd.choice([-4, -1, 0, 1, 4], n) * 2.0 ** -52), 1e-5)
    tl = np.stack([tmin, tmax], axis=1)
    rays = np.concatenate([o, d], axis=1)
    out = np.zeros(n, dtype=np.int32)
    lib = pbe.lib
    lib.rt_debug_aabb_hit.restype = C.c_int
    lib.rt_debug_aabb_hit.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    boxes, rays, tl = (np.ascontiguousarray(x, dtype=np.float64) for x in (boxes, rays, tl))
    assert lib.rt_debug_aabb_hit(n, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, out.ctypes.data) == 0
    use = ((out & 4) != 0) & ((out & 8) != 0)
    assert use.sum() > 0.9 * n
    exact = (out & 1) != 0
    assert np.array_equal(exact[use], ((out >> 1) & 1)[use] != 0)
    assert 0.2 * n < exact[use].sum() < 0.9 * n, "the cases should straddle the boundary"
    culled = use & exact & ((out & 16) == 0)
    assert not culled.any(), f"the f32 filter culled {int(culled.sum())} boxes AABB::hit passes, e.g. case {int(np.flatnonzero(culled)[0])}"
    # the f32 kernels (RT_F32) walk the same filter tree with their leaf test in f32: their filter (margin 7 u) against THAT test
    use32 = (out & 64) != 0
    assert use32.sum() > 0.85 * n
    exact32 = (out & 32) != 0
    assert 0.2 * n < exact32[use32].sum() < 0.9 * n
    culled32 = use32 & exact32 & ((out & 128) == 0)
    assert not culled32.any(), f"the f32 kernels' filter culled {int(culled32.sum())} boxes their exact test passes, e.g. case {int(np.flatnonzero(culled32)[0])}"


def _cube_kat_cases(rnd, n):
    """The hostile cases of the Cube known-answer tests: (boxes n x 6, rays n x 6, [t_min, t_max] n x 2, and which cases are of which kind)."""
    scale = 10.0 ** rnd.uniform(-1, 4, (n, 1))
    lo = rnd.uniform(-1, 1, (n, 3)) * scale * rnd.choice([0.0, 1.0, 30.0], (n, 1), p=[0.2, 0.6, 0.2])
    ext = rnd.uniform(0.05, 1, (n, 3)) * scale
    thin = rnd.integers(0, 6, n) == 0
    ext[thin, rnd.integers(0, 3, thin.sum())] *= 10.0 ** rnd.uniform(-7, -2, thin.sum())
    boxes = np.concatenate([lo, lo + ext], axis=1)
    w = rnd.choice([0.0, 1.0, 0.5, 0.3], (n, 3), p=[0.3, 0.3, 0.2, 0.2])
    tgt = lo + w * ext                                                          # on a face / an edge / a corner / inside
    tgt += rnd.choice([-1.0, 0.0, 1.0], (n, 3)) * 10.0 ** rnd.uniform(-15, -3, (n, 3)) * (np.abs(tgt) + ext)
    kind = rnd.integers(0, 5, n)
    o = tgt + rnd.normal(size=(n, 3)) * scale * 10.0 ** rnd.uniform(-2, 1, (n, 1))     # 0, 1: from outside / anywhere
    on_face = kind == 2                                                         # 2: the origin ON a face (a bounce): leaves outward or inward
    o[on_face] = (lo + rnd.choice([0.0, 1.0], (n, 3)) * ext)[on_face] * rnd.choice([1.0, 1.0 + 2.0 ** -52, 1.0 - 2.0 ** -52], (on_face.sum(), 3)) \
        + (rnd.uniform(0, 1, (n, 3)) * ext * (rnd.integers(0, 2, (n, 3))))[on_face] * 0.0
    k2 = np.flatnonzero(on_face)
    ax = rnd.integers(0, 3, k2.size)
    other = rnd.uniform(0, 1, (k2.size, 3)) * ext[k2] + lo[k2]
    keep = o[k2, ax].copy(); o[k2] = other; o[k2, ax] = keep                    # on the face's plane, somewhere over the face
    inside = kind == 3
    o[inside] = (lo + rnd.uniform(0.01, 0.99, (n, 3)) * ext)[inside]
    d = tgt - o
    rand_dir = (kind == 2) | (kind == 4)
    d[rand_dir] = rnd.normal(size=(rand_dir.sum(), 3))
    d *= 10.0 ** rnd.uniform(-2, 2, (n, 1))
    axis_par = rnd.integers(0, 40, n) == 0
    d[axis_par, rnd.integers(0, 3, axis_par.sum())] = 0.0
    tiny = rnd.integers(0, 50, n) == 1                                           # nearly axis-parallel: plane distances of 1e30 ... 1e45
    d[tiny, rnd.integers(0, 3, tiny.sum())] = rnd.choice([-1.0, 1.0], tiny.sum()) * 10.0 ** rnd.uniform(-44, -28, tiny.sum())
    t_hit = np.linalg.norm(tgt - o, axis=1) / np.maximum(np.linalg.norm(d, axis=1), 1e-300)
    tmax = np.where(rnd.integers(0, 2, n) == 0, np.inf, t_hit * (1.0 + rnd.choice([-4, -1, 0, 1, 4, 1000, 1e6], n) * 2.0 ** -52))
    tmin = np.where(rnd.integers(0, 4, n) == 0, t_hit * (1.0 + rnd.choice([-4, -1, 0, 1, 4], n) * 2.0 ** -52), 1e-5)
    tl = np.stack([tmin, tmax], axis=1)
    rays = np.concatenate([o, d], axis=1)
    return boxes, rays, tl, dict(kind=kind, axis_par=axis_par, tiny=tiny, thin=thin, tmax=tmax, tmin=tmin)


def test_cube_fast_path_against_the_six_rect_tests(pbe, obe):
    """Cube::hit on the device two ways (rt_debug_cube_hit): the reference's six AARect tests in cube.rs:17-24 order under HittableList::hit,
    and the kernels' fast path (rt_kernel.hip: cube_fast — six approximate plane distances, ONE exact rect test for the face that wins).
    Wherever the fast path declares a case CLEAR its answer must be the six tests' answer bit for bit: the same t, the same face, or
    no hit.  Cases aimed at where it could go wrong: rays through points on faces, edges and corners moved by 1e-15 ... 1e-3 of the
    cube, origins on a face (every bounce off a cube), inside the cube, far away; thin and tiny cubes; cubes far from the origin;
    [t_min, t_max] ending within ulps of a hit; axis-parallel rays (zero direction components: never clear)."""
    import ctypes as C
    rnd = np.random.default_rng(23)
    n = 600000
    boxes, rays, tl, about = _cube_kat_cases(rnd, n)
    kind, axis_par, tiny, thin, tmax, tmin = (about[k] for k in ("kind", "axis_par", "tiny", "thin", "tmax", "tmin"))
    out = np.zeros((n, 4))
    lib = pbe.lib
    lib.rt_debug_cube_hit.restype = C.c_int
    lib.rt_debug_cube_hit.argtypes = [C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    boxes, rays, tl = (np.ascontiguousarray(x, dtype=np.float64) for x in (boxes, rays, tl))
    rect_m = float(np.abs(boxes).max()) * 1.0000002
    assert lib.rt_debug_cube_hit(n, rect_m, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, out.ctypes.data) == 0
    t_ref, face_ref, t_fast, code = out[:, 0], out[:, 1].astype(int), out[:, 2], out[:, 3].astype(int)
    clear = (code & 8) != 0
    face_fast = (code & 7) - 1
    assert clear.sum() > 0.5 * n and (~clear).sum() > 0.02 * n, (int(clear.sum()), n)
    hit_ref = ~np.isnan(t_ref)
    assert 0.15 * n < hit_ref.sum() < 0.9 * n
    assert not clear[axis_par].any()
    assert clear[tiny].mean() < 0.5                                             # plane distances beyond 1e36 are never clear (the clamp of [t_min, t_max])
    bad = clear & ((face_fast != face_ref) | (t_fast.view(np.uint64) != t_ref.view(np.uint64)) & ~(np.isnan(t_fast) & np.isnan(t_ref)))
    assert not bad.any(), f"{int(bad.sum())} clear cases differ from the six rect tests, e.g. case {int(np.flatnonzero(bad)[0])}: " \
                          f"ref (t {t_ref[np.flatnonzero(bad)[0]]!r}, face {face_ref[np.flatnonzero(bad)[0]]}) fast (t {t_fast[np.flatnonzero(bad)[0]]!r}, face {face_fast[np.flatnonzero(bad)[0]]})"
    # ... and the device's six-test side IS the reference's Cube::hit: the oracle's (cube.rs:14-37 over rect.rs:49-60), same t bit for
    # bit, same face, on every one of the cases (round 6: until then the known-answer test was device against device)
    ref2 = np.zeros((n, 2))
    obe.lib.orc_cube_hit_batch(n, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, ref2.ctypes.data)
    t_orc, face_orc = ref2[:, 0], ref2[:, 1].astype(int)
    assert (face_orc >= -1).all()
    differ = (face_orc != face_ref) | ((t_orc.view(np.uint64) != t_ref.view(np.uint64)) & ~(np.isnan(t_orc) & np.isnan(t_ref)))
    assert not differ.any(), f"{int(differ.sum())} cases: the device's six rect tests differ from the oracle's Cube::hit, e.g. case {int(np.flatnonzero(differ)[0])}"
    # the same cubes with the scene-wide bound far larger than their own coordinates (other rects of the scene): still exact
    assert lib.rt_debug_cube_hit(n, rect_m * 1000.0, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, out.ctypes.data) == 0
    clear2 = (out[:, 3].astype(int) & 8) != 0
    bad2 = clear2 & (((out[:, 3].astype(int) & 7) - 1 != out[:, 1].astype(int)) | (out[:, 2].view(np.uint64) != out[:, 0].view(np.uint64)) & ~(np.isnan(out[:, 2]) & np.isnan(out[:, 0])))
    assert not bad2.any()
    # plain rays are (nearly) always clear: the fast path is what runs
    plain = (kind == 4) & ~axis_par & ~tiny & ~thin & (np.isinf(tmax)) & (tmin == 1e-5)   # a random direction from a random origin
    assert clear[plain].mean() > 0.99, clear[plain].mean()


def test_room_form_of_the_cube_fast_path_against_the_walls_rect_tests(pbe, obe):
    """The fast path's ROOM form (round 6: walls of a list scene that are faces of one box, rt_flatten.cpp form_room — the Cornell room's five
    walls): the same hostile cases as the Cube test, each with a random set of faces that exist (every set of 0 ... 6).  Wherever the fast
    path declares a case CLEAR its answer must be HittableList::hit's over the rects of the faces that exist, bit for bit — the entry face
    if it exists and is in range, else the exit face if it exists and is in range, else no hit — and the device's exact side IS the
    oracle's list over those AARects (orc_room_hit_batch) on every case.  With all six faces the room form must say what the Cube form says."""
    import ctypes as C
    rnd = np.random.default_rng(29)
    n = 600000
    boxes, rays, tl, about = _cube_kat_cases(rnd, n)
    masks = rnd.integers(0, 64, n).astype(np.uint32)
    # (bit f = face f of cube.rs:17-24: z max, z min, y max, y min, x max, x min)
    masks[rnd.integers(0, 5, n) == 0] = 0x3D                                    # the Cornell room of main.rs:291-296: everything but the face at min z, the open front
    masks[rnd.integers(0, 10, n) == 0] = 0x3E                                   # ... but the one at max z
    masks[rnd.integers(0, 20, n) == 0] = 0x3F                                   # all six
    out = np.zeros((n, 4))
    lib = pbe.lib
    lib.rt_debug_room_hit.restype = C.c_int
    lib.rt_debug_room_hit.argtypes = [C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    boxes, rays, tl = (np.ascontiguousarray(x, dtype=np.float64) for x in (boxes, rays, tl))
    rect_m = float(np.abs(boxes).max()) * 1.0000002
    assert lib.rt_debug_room_hit(n, rect_m, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, masks.ctypes.data, out.ctypes.data) == 0
    t_ref, face_ref, t_fast, code = out[:, 0].copy(), out[:, 1].astype(int), out[:, 2].copy(), out[:, 3].astype(int)
    clear = (code & 8) != 0
    face_fast = (code & 7) - 1
    assert clear.sum() > 0.5 * n and (~clear).sum() > 0.02 * n, (int(clear.sum()), n)
    hit_ref = ~np.isnan(t_ref)
    assert 0.05 * n < hit_ref.sum() < 0.9 * n
    assert not clear[about["axis_par"]].any()
    assert (((masks[hit_ref] >> face_ref[hit_ref]) & 1) == 1).all(), "a face that does not exist was hit"
    bad = clear & ((face_fast != face_ref) | (t_fast.view(np.uint64) != t_ref.view(np.uint64)) & ~(np.isnan(t_fast) & np.isnan(t_ref)))
    assert not bad.any(), f"{int(bad.sum())} clear cases differ from the walls' rect tests, e.g. case {int(np.flatnonzero(bad)[0])}: mask {int(masks[np.flatnonzero(bad)[0]]):06b} " \
                          f"ref (t {t_ref[np.flatnonzero(bad)[0]]!r}, face {face_ref[np.flatnonzero(bad)[0]]}) fast (t {t_fast[np.flatnonzero(bad)[0]]!r}, face {face_fast[np.flatnonzero(bad)[0]]})"
    # the device's exact side is the oracle's HittableList over the AARects of the faces that exist
    ref2 = np.zeros((n, 2))
    obe.lib.orc_room_hit_batch(n, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, masks.ctypes.data, ref2.ctypes.data)
    t_orc, face_orc = ref2[:, 0], ref2[:, 1].astype(int)
    assert (face_orc >= -1).all()
    differ = (face_orc != face_ref) | ((t_orc.view(np.uint64) != t_ref.view(np.uint64)) & ~(np.isnan(t_orc) & np.isnan(t_ref)))
    assert not differ.any(), f"{int(differ.sum())} cases: the device's wall tests differ from the oracle's list, e.g. case {int(np.flatnonzero(differ)[0])}"
    # an absent face matters: among the clear cases some hit the exit face BECAUSE the entry face does not exist, some nothing at all
    full = np.zeros((n, 4))
    lib.rt_debug_cube_hit.restype = C.c_int
    lib.rt_debug_cube_hit.argtypes = [C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.rt_debug_cube_hit(n, rect_m, boxes.ctypes.data, rays.ctypes.data, tl.ctypes.data, full.ctypes.data) == 0
    cube_face = (full[:, 3].astype(int) & 7) - 1
    cube_clear = (full[:, 3].astype(int) & 8) != 0
    both = clear & cube_clear
    assert (both & (cube_face >= 0) & (face_fast >= 0) & (cube_face != face_fast)).sum() > 1000, "no case where the exit face stands in for an absent entry face"
    assert (both & (cube_face >= 0) & (face_fast < 0)).sum() > 1000
    # every face there: the room form is the Cube form
    six = masks == 0x3F
    assert six.sum() > 5000
    same = (code[six] == full[six, 3].astype(int)) & ((t_fast[six].view(np.uint64) == full[six, 2].view(np.uint64)) | (np.isnan(t_fast[six]) & np.isnan(full[six, 2])))
    assert same.all()
    # plain rays are (nearly) always clear with any set of faces
    plain = (about["kind"] == 4) & ~about["axis_par"] & ~about["tiny"] & ~about["thin"] & np.isinf(about["tmax"]) & (about["tmin"] == 1e-5)
    assert clear[plain].mean() > 0.99, clear[plain].mean()


@pytest.mark.parametrize("name", ["random", "final", "mesh0", "teapot"])
def test_speculative_box_steps_are_scheduling_only(name, pbe, obe, orc_mod, earth):
    """RT_SPECULATE_BVH: in the lock-step BVH walk a lane that has reached a leaf walks on along the leaf's skip link while it waits for
    the leaf step, holds at most two leaves, and a leaf reached past an untested one is re-tested against its own box with the closest hit
    as it is when its turn comes (rt_kernel.hip: bvh_hit_filt<.., SPEC>; exact by containment of a child's slab interval in its ancestors' —
    since round 5 EVERY leaf is re-tested against its own box in the leaf step, which is what makes walking ahead free).
    Chosen automatically when the world is ONE BVH (random spheres); forced on here for the others.  Every sample is bit-identical to the
    plain walk and matches the oracle."""
    mk = (lambda be: _mesh_room(be, 0)) if name == "mesh0" else (lambda be: build_scene(name, be, earth))
    b, cam, bg = mk(pbe)
    W, H, spp, depth = 96, 54, 8, 30
    _, plain = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_NO_SPECULATE_BVH, want_samples=True)
    _, spec = R.render(b, cam, bg, W, H, spp, depth, flags=R.RT_LOCKSTEP_BVH | R.RT_SPECULATE_BVH, want_samples=True)
    assert np.array_equal(plain.view(np.uint64), spec.view(np.uint64))
    _, auto = R.render(b, cam, bg, W, H, spp, depth, want_samples=True)
    assert np.array_equal(plain.view(np.uint64), auto.view(np.uint64))
    ob, ocam, obg = mk(obe)
    _, ref = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True)
    n_div, _, _ = _compare_samples(spec, ref)
    assert n_div <= MAX_DIVERGED


def test_rotation_by_an_angle_whose_sincos_differs_from_sin_and_cos(pbe, obe, orc_mod):
    """Found by the round-3 fuzz sweep (seed 38793: 35 diverged samples in one scene).  glibc's sincos() differs from its sin() / cos() in
    the last ulp for 0.13 % of arguments — one of them (pi/180) * -73.5789378649801 — and LLVM (hence rustc) turns `radians.sin()` /
    `radians.cos()` of Rotate::new (rotate.rs:35-36) into one sincos call, as g++ -O3 does in the oracle, while the product's flattener
    (clang) made two calls: the rotation matrices differed by an ulp, and behind such a Rotate two coincident triangles (a Mesh that names
    one triangle twice with its vertices rotated: their t differ in the last ulp) swapped which one wins.  Both sides call sincos explicitly
    now; this is that scene's mesh: every sample must be bit-comparable again."""
    def build(be):
        b = SceneBuilder(be)
        glow = b.DiffuseLight(b.ConstantTexture((4.0, 4.0, 4.0)))
        grey = b.Lambertian(b.ConstantTexture((0.6, 0.5, 0.4)))
        verts = [(24.623369398270484, 39.450005170621395, 9.599462891678257), (17.623561837599567, -38.37418967943361, 12.201314173323262),
                 (39.8930862122764, -18.061201982550102, 26.50247531565779), (39.724616296882246, 23.61365762872029, 23.618362277751473),
                 (19.023481207843886, -3.657442987881069, -29.076281302200968)]
        mesh = b.Mesh(verts, [3, 2, 1, 1, 3, 2, 2, 3, 4], grey)
        h = b.Rotate(Axis.Y, b.Rotate(Axis.Y, b.Rotate(Axis.Y, mesh, 9.042999908477967), 27.074643535137966), -73.5789378649801)
        world = b.HittableList()
        world.push(h)
        lamp = b.AARect(Plane.XZ, -60.0, 60.0, -60.0, 60.0, 90.0, glow)
        world.push(lamp)
        b.set_scene(world, [lamp])
        return b, Camera((24.0, 13.0, -170.0), (17.0, 16.0, 42.0), (0.0, 1.0, 0.0), 25.0, 1.0, 0.0, 10.0, 0.0, 1.0), (0.1, 0.1, 0.2)
    pb, pcam, pbg = build(pbe)
    ob, ocam, obg = build(obe)
    W, H, spp, depth = 64, 64, 8, 10
    _, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True)
    _, rs = orc_mod.render(ob, ocam, obg, W, H, spp, depth, want_samples=True)
    n_bad, max_d, _ = _compare_samples(gs, rs)
    assert (rs.sum(axis=-1) > 0).mean() > 0.05           # the mesh is in view and lit
    assert n_bad == 0, f"{n_bad} samples diverged (max |d| of the rest {max_d:.2e})"
