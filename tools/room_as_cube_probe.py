"""What would the Cornell room's walls cost as ONE Cube through the shipped fast path (cube_fast: one exact rect test instead of six)?
An upper bound measured WITHOUT any kernel change, by a scene-level A/B in one process (interleaved rounds, min kernel ms):

    A  a closed room of SIX wall rects (the five of main.rs:291-296 plus the missing front wall), the light, the two boxes
    B  the same room as ONE Cube (0,0,0)-(555,555,555), the light, the two boxes

camera INSIDE the room (a Cube has six faces: the reference's camera outside the open front would see nothing), every wall white in
both (albedo only scales the throughput; nothing here ends a path by its colour), so both scenes trace the same paths bounce for
bounce and the difference in time is six exact rect tests against one Cube test.  A masked five-face room saves less:
(5 x 104 - 330) / (6 x 104 - 330) = 0.65 of it by the section costs of DESIGN.md 3.2.

usage: python tools/room_as_cube_probe.py [--spp 256] [--rounds 5]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime in the process)
from raytracinginrust_amd import _lib
from raytracinginrust_amd.api import SceneBuilder, Camera, Plane, Axis

ap = argparse.ArgumentParser()
ap.add_argument('--spp', type=int, default=256); ap.add_argument('--rounds', type=int, default=5); ap.add_argument('--size', type=int, default=800)
a = ap.parse_args()
be = _lib.load()


def room(as_cube):
    b = SceneBuilder(be)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    metal = b.Metal((0.8, 0.85, 0.88), 0.0)
    light = b.DiffuseLight(b.ConstantTexture((15.0, 15.0, 15.0)))
    rect_light = b.FlipNormal(b.AARect(Plane.XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light))
    world = b.HittableList()
    if as_cube:
        world.push(b.Cube((0.0, 0.0, 0.0), (555.0, 555.0, 555.0), white))
        world.push(rect_light)
    else:
        world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, white))
        world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
        world.push(rect_light)
        world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
        world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white))
        world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
        world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white), -18.0), (130.0, 0.0, 65.0)))
    world.push(b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), metal), 15.0), (265.0, 0.0, 295.0)))
    b.set_scene(world, [rect_light])
    cam = Camera((278.0, 278.0, 5.0), (278.0, 278.0, 555.0), (0.0, 1.0, 0.0), 80.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


variants = [("six_rects", *room(False)), ("one_cube", *room(True))]
W = H = a.size


def run(v):
    name, b, cam, bg = v
    out = np.zeros((H, W, 3))
    rc = be.lib.rt_render(b.h, C.byref(cam), (C.c_double * 3)(*bg), W, H, a.spp, 50, 0x5EED, 0, out.ctypes.data)
    assert rc == 0, be.lib.rt_last_error()
    ms = C.c_float(); be.lib.rt_last_kernel_ms(b.h, C.byref(ms))
    st = (C.c_ulonglong * 3)(); be.lib.rt_last_stats(b.h, st)
    return out, ms.value, int(st[0])


ref = None; times = {v[0]: [] for v in variants}; iters = {}
for r in range(a.rounds + 1):
    for v in variants:
        out, ms, it = run(v)
        iters[v[0]] = it
        if r == 0:
            if ref is None: ref = out
            else:
                d = np.abs(out - ref)
                print(f'{v[0]}: max |diff| vs {variants[0][0]} = {np.nanmax(d):.3e}; frame means {np.nanmean(ref) / a.spp:.6f} / {np.nanmean(out) / a.spp:.6f}; '
                      f'{float((out.view(np.uint64) == ref.view(np.uint64)).mean()):.1%} of the sums bit-identical')
        else: times[v[0]].append(ms)
for name, ts in times.items():
    ts = sorted(ts); print(f'{name:12s} min {ts[0]:8.3f} ms  median {ts[len(ts)//2]:8.3f} ms  {W*H*a.spp/ts[0]/1e3:9.1f} Msamples/s  wave bounce-iterations {iters[name]}')
t6, t1 = min(times["six_rects"]), min(times["one_cube"])
print(f'six rects -> one Cube: {100.0 * (t6 / t1 - 1.0):+.1f} % (upper bound; a five-face masked room: about {100.0 * 0.65 * (t6 / t1 - 1.0):+.1f} %)')
