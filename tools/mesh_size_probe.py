"""Developer probe: persistent-traversal loop vs lock-step loop on a lit room with two triangle-mesh BVHs of n triangles each (the rule that
picks the persistent loop for mesh scenes was tuned on the teapot room).   usage: python tools/mesh_size_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from raytracinginrust_amd import _lib, render as R
from raytracinginrust_amd.api import Axis, Camera, Plane, SceneBuilder
be = _lib.load()
def room(n, two=True):
    rs = np.random.RandomState(n)
    b = SceneBuilder(be)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73))); red = b.Lambertian(b.ConstantTexture((0.65, 0.05, 0.05)))
    steel = b.Metal((0.8, 0.85, 0.88), 0.1); light = b.DiffuseLight(b.ConstantTexture((12.0, 12.0, 12.0)))
    world = b.HittableList()
    lamp = b.FlipNormal(b.AARect(Plane.XZ, 150.0, 400.0, 150.0, 400.0, 554.0, light)); world.push(lamp)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white)); world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, red))
    def blob(center, radius, n, mat):
        tris = []
        s = 25.0 * (100.0 / max(n, 100)) ** 0.5
        for _ in range(n):
            p0 = center + rs.uniform(-radius, radius, 3)
            tris.append(b.Triangle([tuple(p0), tuple(p0 + rs.uniform(-s, s, 3)), tuple(p0 + rs.uniform(-s, s, 3))], mat))
        return b.BVH(tris, 0.0, 1.0)
    world.push(blob(np.array([200.0, 120.0, 250.0]), 70.0, n, white))
    if two: world.push(b.Translate(b.Rotate(Axis.Y, blob(np.array([0.0, 0.0, 0.0]), 60.0, n, steel), 25.0), (380.0, 200.0, 300.0)))
    b.set_scene(world, [lamp])
    return b, Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0), (0.02, 0.02, 0.03)
for two in (True, False):
    for n in (32, 100, 300, 1000, 3000):
        b, cam, bg = room(n, two)
        res = {}
        for tag, fl in (('persistent', R.RT_PERSISTENT_BVH), ('lockstep', R.RT_LOCKSTEP_BVH)):
            ms = []
            for _ in range(3):
                R.render(b, cam, bg, 400, 400, 64, 50, flags=fl); ms.append(R.last_kernel_ms(b))
            res[tag] = min(ms)
        print(f'{"two trees" if two else "one tree "} of {n:5d} triangles ({R.flatten(b)["bvh_nodes"]} nodes): persistent {res["persistent"]:7.3f} ms, lock-step {res["lockstep"]:7.3f} ms, ratio {res["persistent"] / res["lockstep"]:.2f}', flush=True)
