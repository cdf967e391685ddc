"""What the F_NESTED instantiation costs: the random-spheres scene (scenes.random_scene, BASELINE config 1's scene) with its 533 spheres
in ONE BVH (the reference's scene: the walk-ahead kernel <double, 2111u>) and the same spheres as a BVH of 23 row-BVHs (BVHs as BVH
children: sub-object leaves, the <double, 639u> kernel, two levels of walk).  Same closest hits either way — the frames agree to the
tolerance of a tie — so the ratio of the kernel times is the price of the general form.   usage: python tools/nested_probe.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R, scenes
from raytracinginrust_amd.api import Camera, Rng, SceneBuilder
be = _lib.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def build(nested):
    b = SceneBuilder(be)
    rng = Rng(be, scenes.DEFAULT_SEED, scenes.STREAM_RANDOM_SCENE)
    ground = b.Sphere((0.0, -1000.0, 0.0), 1000.0, b.Lambertian(b.CheckTexture(b.ConstantTexture((1.0, 1.0, 1.0)), b.ConstantTexture((0.3, 0.3, 1.0)))))
    rows = []
    for a in range(-11, 12):
        row = []
        for bb in range(-11, 12):
            choose = rng.gen_f64()
            center = (float(a) + rng.gen_range(0.0, 0.9), 0.2, float(bb) + rng.gen_range(0.0, 0.9))
            if choose < 0.8:
                c1, c2 = rng.color_random(0.0, 1.0), rng.color_random(0.0, 1.0)
                mat = b.Lambertian(b.ConstantTexture((c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2])))
                row.append(b.MovingSphere(center, scenes.add(center, (0.0, rng.gen_range(0.0, 0.01), 0.0)), 0.0, 1.0, 0.2, mat))
            elif choose < 0.95:
                albedo = rng.color_random(0.4, 1.0)
                row.append(b.Sphere(center, 0.2, b.Metal(albedo, rng.gen_range(0.0, 0.5))))
            else:
                row.append(b.Sphere(center, 0.2, b.Dielectric(1.5)))
        rows.append(row)
    big = [b.Sphere((0.0, 1.0, 0.0), 1.0, b.Dielectric(1.5)), b.Sphere((-4.0, 1.0, 0.0), 1.0, b.Lambertian(b.ConstantTexture((0.4, 0.2, 0.1)))),
           b.Sphere((4.0, 1.0, 0.0), 1.0, b.Metal((0.7, 0.6, 0.5), 0.0))]
    if nested:
        world = b.BVH([ground] + [b.BVH(r, 0.0, 1.0) for r in rows] + big, 0.0, 1.0)
    else:
        world = b.BVH([ground] + [s for r in rows for s in r] + big, 0.0, 1.0)
    b.set_scene(world, [])
    return b, Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, 1.0, 0.1, 10.0, 0.0, 1.0), (0.7, 0.8, 1.0)


W = H = 800
res = {}
for name, nested in (("one BVH", False), ("BVH of 23 row-BVHs", True)):
    b, cam, bg = build(nested)
    ms = []
    for _ in range(4):
        img = R.render(b, cam, bg, W, H, spp, 50)
        ms.append(R.last_kernel_ms(b))
    li = R.last_loop_info(b)
    res[name] = (min(ms[1:]), img, li)
    print(f"{name:22s} {li['kernel']:40s} kernel {min(ms[1:]):8.2f} ms  {W * H * spp / min(ms[1:]) / 1e3:8.1f} Msamples/s  frame mean {np.nanmean(img) / spp:.5f}")
a, c = res["one BVH"][1], res["BVH of 23 row-BVHs"][1]
print(f"frames: max |difference of pixel sums| {np.nanmax(np.abs(a - c)):.3e} over sums up to {np.nanmax(np.abs(a)):.1f} (equal-t ties resolve by tree order; everything else is the same hit)")
print(f"nested / flat kernel time: {res['BVH of 23 row-BVHs'][0] / res['one BVH'][0]:.2f}")
