"""The reference's scene functions and cameras (src/main.rs), restated over `SceneBuilder`.

Each function returns `(builder, camera, background)`; the scene's `(world, lights)` pair is
already attached to the builder.  Random draws (`rand::thread_rng()` in the reference) come from
a seeded stream `Rng(backend, seed, stream)` and keep the source order of the reference's draws.
All ten scene functions of the reference are here (`Scene` enum, src/main.rs:564-575).
"""
from __future__ import annotations

import os
import struct

from .api import Axis, Backend, Camera, Plane, Rng, SceneBuilder

# host-side stream ids (pixel index 0xFFFFFFFF is reserved for them, see csrc/rt_rng.h)
STREAM_RANDOM_SCENE = 0
STREAM_FINAL_SCENE = 1
DEFAULT_SEED = 0x5EED


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def random_scene(backend: Backend, seed: int = DEFAULT_SEED, aspect_ratio: float = 16.0 / 9.0):
    """src/main.rs:153-210, camera src/main.rs:628-635.  `lights` is empty."""
    b = SceneBuilder(backend)
    rng = Rng(backend, seed, STREAM_RANDOM_SCENE)
    world = []
    ground_mat = b.Lambertian(b.CheckTexture(b.ConstantTexture((1.0, 1.0, 1.0)), b.ConstantTexture((0.3, 0.3, 1.0))))
    world.append(b.Sphere((0.0, -1000.0, 0.0), 1000.0, ground_mat))
    for a in range(-11, 12):
        for bb in range(-11, 12):
            choose_mat = rng.gen_f64()
            cx = float(a) + rng.gen_range(0.0, 0.9)
            cz = float(bb) + rng.gen_range(0.0, 0.9)
            center = (cx, 0.2, cz)
            if choose_mat < 0.8:
                c1 = rng.color_random(0.0, 1.0)
                c2 = rng.color_random(0.0, 1.0)
                albedo = (c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2])
                mat = b.Lambertian(b.ConstantTexture(albedo))
                center1 = add(center, (0.0, rng.gen_range(0.0, 0.01), 0.0))
                world.append(b.MovingSphere(center, center1, 0.0, 1.0, 0.2, mat))
            elif choose_mat < 0.95:
                albedo = rng.color_random(0.4, 1.0)
                fuzz = rng.gen_range(0.0, 0.5)
                world.append(b.Sphere(center, 0.2, b.Metal(albedo, fuzz)))
            else:
                world.append(b.Sphere(center, 0.2, b.Dielectric(1.5)))
    world.append(b.Sphere((0.0, 1.0, 0.0), 1.0, b.Dielectric(1.5)))
    world.append(b.Sphere((-4.0, 1.0, 0.0), 1.0, b.Lambertian(b.ConstantTexture((0.4, 0.2, 0.1)))))
    world.append(b.Sphere((4.0, 1.0, 0.0), 1.0, b.Metal((0.7, 0.6, 0.5), 0.0)))
    b.set_scene(b.BVH(world, 0.0, 1.0), [])
    cam = Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect_ratio, 0.1, 10.0, 0.0, 1.0)
    return b, cam, (0.7, 0.8, 1.0)


def two_spehre(backend: Backend, aspect_ratio: float = 16.0 / 9.0):
    """src/main.rs:212-227 (name as in the reference), camera src/main.rs:640-649.  `lights` empty."""
    b = SceneBuilder(backend)
    world = b.HittableList()
    top_mat = b.Lambertian(b.CheckTexture(b.ConstantTexture((1.0, 1.0, 1.0)), b.ConstantTexture((0.3, 0.3, 1.0))))
    bottom_mat = b.Lambertian(b.CheckTexture(b.ConstantTexture((1.0, 1.0, 1.0)), b.ConstantTexture((0.3, 0.3, 1.0))))
    world.push(b.Sphere((0.0, 10.0, 0.0), 10.0, top_mat))
    world.push(b.Sphere((0.0, -10.0, 0.0), 10.0, bottom_mat))
    b.set_scene(world, [])
    cam = Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.7, 0.8, 1.0)


STREAM_TWO_PERLIN = 2


def two_perlin_sphere(backend: Backend, seed: int = DEFAULT_SEED, aspect_ratio: float = 16.0 / 9.0):
    """src/main.rs:229-245 (objects moved to the first quadrant because of the negative-lattice quirk, :235),
    camera src/main.rs:654-663.  Two separate NoiseTexture::new(2.0), i.e. two Perlin tables, top first."""
    b = SceneBuilder(backend)
    rng = Rng(backend, seed, STREAM_TWO_PERLIN)
    world = b.HittableList()
    top_mat = b.Lambertian(b.NoiseTexture(2.0, rng))
    bottom_mat = b.Lambertian(b.NoiseTexture(2.0, rng))
    world.push(b.Sphere((1000.0, 2.0, 1000.0), 2.0, top_mat))
    world.push(b.Sphere((1000.0, -1000.0, 1000.0), 1000.0, bottom_mat))
    b.set_scene(world, [])
    cam = Camera((1013.0, 2.0, 1003.0), (1000.0, 0.0, 1000.0), (0.0, 1.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.7, 0.8, 1.0)


def earth(backend: Backend, earth_rgb8: bytes, earth_w: int, earth_h: int, aspect_ratio: float = 16.0 / 9.0):
    """src/main.rs:247-255 (world is the bare sphere, not a list), camera src/main.rs:668-677."""
    b = SceneBuilder(backend)
    globe = b.Sphere((0.0, 0.0, 0.0), 2.0, b.Lambertian(b.ImageTexture(earth_rgb8, earth_w, earth_h)))
    b.set_scene(globe, [])
    cam = Camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect_ratio, 0.1, 10.0, 0.0, 1.0)
    return b, cam, (0.7, 0.8, 1.0)


def light_room(backend: Backend, aspect_ratio: float = 16.0 / 9.0):
    """src/main.rs:257-276, camera src/main.rs:682-691.  The XY rect light is pushed un-flipped."""
    b = SceneBuilder(backend)
    world = b.HittableList()
    bottom_mat = b.Lambertian(b.ConstantTexture((0.7, 0.7, 0.7)))
    top_mat = b.Lambertian(b.ConstantTexture((0.0, 0.1843, 0.6549)))
    emitted = b.DiffuseLight(b.ConstantTexture((4.0, 4.0, 4.0)))
    world.push(b.Sphere((0.0, -1000.0, 0.0), 1000.0, bottom_mat))
    world.push(b.Sphere((0.0, 2.0, 0.0), 2.0, top_mat))
    plane = b.AARect(Plane.XY, 3.0, 5.0, 1.0, 3.0, -2.0, emitted)
    world.push(plane)
    b.set_scene(world, [plane])
    cam = Camera((26.0, 3.0, 6.0), (0.0, 2.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def cornell_box_with_smoke(backend: Backend, aspect_ratio: float = 1.0):
    """src/main.rs:313-346, camera src/main.rs:712-719.  With the committed code the media absorb;
    the RT_ISOTROPIC_SCATTER render flag gives the scattering look of img/volume.png."""
    b = SceneBuilder(backend)
    red = b.Lambertian(b.ConstantTexture((0.65, 0.05, 0.05)))
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    green = b.Lambertian(b.ConstantTexture((0.12, 0.45, 0.15)))
    light = b.DiffuseLight(b.ConstantTexture((15.0, 15.0, 15.0)))
    rect_light = b.FlipNormal(b.AARect(Plane.XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light))
    world = b.HittableList()
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green))
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red))
    world.push(rect_light)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    box1 = b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white), -18.0), (130.0, 0.0, 65.0))
    box2 = b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white), 15.0), (265.0, 0.0, 295.0))
    world.push(b.ConstantMedium(box1, 0.01, b.ConstantTexture((1.0, 1.0, 1.0))))
    world.push(b.ConstantMedium(box2, 0.01, b.ConstantTexture((0.0, 0.0, 0.0))))
    b.set_scene(world, [rect_light])
    cam = Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, aspect_ratio, 0.05, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def progress_showcase(backend: Backend, aspect_ratio: float = 1.0):
    """src/main.rs:515-562: every line of the body is commented out, so (world, lights) are two empty lists.
    Camera src/main.rs:754-761.  Every ray misses: the frame is the background."""
    b = SceneBuilder(backend)
    b.set_scene(b.HittableList(), [])
    cam = Camera((-3.3, 6.8, -9.8), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 40.0, aspect_ratio, 0.2, 12.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def cornell_box(backend: Backend, aspect_ratio: float = 1.0):
    """src/main.rs:278-311, camera src/main.rs:698-705."""
    b = SceneBuilder(backend)
    red = b.Lambertian(b.ConstantTexture((0.65, 0.05, 0.05)))
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    green = b.Lambertian(b.ConstantTexture((0.12, 0.45, 0.15)))
    metal = b.Metal((0.8, 0.85, 0.88), 0.0)
    light = b.DiffuseLight(b.ConstantTexture((15.0, 15.0, 15.0)))
    rect_light = b.FlipNormal(b.AARect(Plane.XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light))
    world = b.HittableList()
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green))
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red))
    world.push(rect_light)
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white), -18.0), (130.0, 0.0, 65.0)))
    world.push(b.Translate(b.Rotate(Axis.Y, b.Cube((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), metal), 15.0), (265.0, 0.0, 295.0)))
    b.set_scene(world, [rect_light])
    cam = Camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, aspect_ratio, 0.05, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def final_scene(backend: Backend, earth_rgb8: bytes, earth_w: int, earth_h: int, seed: int = DEFAULT_SEED, aspect_ratio: float = 1.0):
    """src/main.rs:453-513, camera src/main.rs:740-747.  `earth_rgb8` is the decoded earthmap (ImageTexture::new)."""
    b = SceneBuilder(backend)
    rng = Rng(backend, seed, STREAM_FINAL_SCENE)
    world = b.HittableList()
    ground = b.Lambertian(b.ConstantTexture((0.48, 0.83, 0.53)))
    box_list1 = []
    boxes_per_side = 20
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            w = 100.0
            x0 = -1000.0 + float(i) * w
            z0 = -1000.0 + float(j) * w
            y0 = 0.0
            x1 = x0 + w
            y1 = 100.0 * (rng.gen_f64() + 0.01)
            z1 = z0 + w
            box_list1.append(b.Cube((x0, y0, z0), (x1, y1, z1), ground))
    world.push(b.BVH(box_list1, 0.0, 1.0))
    light = b.DiffuseLight(b.ConstantTexture((7.0, 7.0, 7.0)))
    rect_light = b.FlipNormal(b.AARect(Plane.XZ, 147.0, 412.0, 123.0, 423.0, 554.0, light))
    world.push(rect_light)
    center = (400.0, 400.0, 200.0)
    world.push(b.MovingSphere(center, add(center, (30.0, 0.0, 0.0)), 0.0, 1.0, 50.0, b.Lambertian(b.ConstantTexture((0.7, 0.3, 0.1)))))
    world.push(b.Sphere((260.0, 150.0, 45.0), 50.0, b.Dielectric(1.5)))
    world.push(b.Sphere((0.0, 150.0, 145.0), 50.0, b.Metal((0.8, 0.8, 0.9), 1.0)))
    boundary = b.Sphere((360.0, 150.0, 145.0), 70.0, b.Dielectric(1.5))
    world.push(boundary)
    world.push(b.ConstantMedium(boundary, 0.2, b.ConstantTexture((0.2, 0.4, 0.9))))
    boundary = b.Sphere((0.0, 0.0, 0.0), 5000.0, b.Dielectric(1.5))
    world.push(b.ConstantMedium(boundary, 0.0001, b.ConstantTexture((1.0, 1.0, 1.0))))
    world.push(b.Sphere((400.0, 200.0, 400.0), 100.0, b.Lambertian(b.ImageTexture(earth_rgb8, earth_w, earth_h))))
    world.push(b.Sphere((220.0, 280.0, 300.0), 80.0, b.Lambertian(b.NoiseTexture(0.1, rng))))
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    box_list2 = []
    for _ in range(1000):
        x = 165.0 * rng.gen_f64()
        y = 165.0 * rng.gen_f64()
        z = 165.0 * rng.gen_f64()
        box_list2.append(b.Sphere((x, y, z), 10.0, white))
    world.push(b.Translate(b.Rotate(Axis.Y, b.BVH(box_list2, 0.0, 0.1), 15.0), (-100.0, 270.0, 395.0)))
    b.set_scene(world, [rect_light])
    cam = Camera((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, aspect_ratio, 0.01, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def load_obj(path: str, offset, scale: float):
    """Mesh::load_obj's parsing step (src/mesh.rs:33-61) with tobj 3.2.3 semantics for what the path uses:
    positions parsed as f32 then widened (`p[0] as f64`), faces fan-triangulated, only the first model,
    then `* scale + offset` in f64.  Returns (positions, indices)."""
    pos, idx = [], []
    have_faces = False
    with open(path, "r") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                pos.append(tuple(struct.unpack("f", struct.pack("f", float(x)))[0] for x in t[1:4]))
            elif t[0] == "f":
                have_faces = True
                vs = []
                for tok in t[1:]:
                    k = int(tok.split("/")[0])
                    vs.append(k - 1 if k > 0 else len(pos) + k)
                for k in range(1, len(vs) - 1):
                    idx += [vs[0], vs[k], vs[k + 1]]
            elif t[0] in ("o", "g") and have_faces:
                break   # models[0] only (src/mesh.rs:42)
    positions = [(p[0] * scale + offset[0], p[1] * scale + offset[1], p[2] * scale + offset[2]) for p in pos]
    return positions, idx


# C4: the reference never places teapot.obj (its scene loads the absent Venus.obj, src/main.rs:431, a tall
# statue the camera at src/main.rs:728-729 looks at around y = 375).  This placement is ours: the teapot
# (bbox x[-84.4,98.3] y[-39.9,49.7] z[-58.0,55.7]) scaled 1.5 and centred near that look-at point.
TEAPOT_SCALE = 1.5
TEAPOT_OFFSET = (268.0, 340.0, 258.0)


def cornell_test(backend: Backend, obj_path: str, aspect_ratio: float = 1.0, scale: float = TEAPOT_SCALE, offset=TEAPOT_OFFSET):
    """src/main.rs:348-451 as committed (walls, one light, BVH of the mesh), camera src/main.rs:726-733."""
    b = SceneBuilder(backend)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    desire = b.Lambertian(b.ConstantTexture((0.922, 0.238, 0.331)))
    safety_orange = b.Lambertian(b.ConstantTexture((1.000, 0.471, 0.0)))
    color_80cf00 = b.Lambertian(b.ConstantTexture((0.502, 0.812, 0.002)))
    light0 = b.DiffuseLight(b.ConstantTexture((1.0 * 2.2, 1.0 * 2.2, 0.88 * 2.2)))
    world = b.HittableList()
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 555.0, desire))
    world.push(b.AARect(Plane.YZ, 0.0, 555.0, 0.0, 555.0, 0.0, safety_orange))
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white))
    world.push(b.AARect(Plane.XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    world.push(b.AARect(Plane.XY, 0.0, 555.0, 0.0, 555.0, 555.0, white))
    rect_light0 = b.FlipNormal(b.AARect(Plane.XZ, 128.0, 428.0, 115.0, 270.0, 554.0, light0))
    positions, indices = load_obj(obj_path, offset, scale)
    obj_tris = b.Mesh(positions, indices, color_80cf00)
    world.push(rect_light0)
    world.push(b.BVH(obj_tris, 0.0, 1.0))
    b.set_scene(world, [rect_light0])
    cam = Camera((199.0, 439.0, -200.0), (278.0, 375.0, 258.0), (0.0, 1.0, 0.0), 30.0, aspect_ratio, 0.01, 10.0, 0.0, 1.0)
    return b, cam, (0.0, 0.0, 0.0)


def load_image_rgb8(path: str):
    """`image::open(path).to_rgb8()` (src/main.rs:248,491) -> (bytes, width, height).  JPEG goes through the library's own
    baseline decoder (csrc/rt_jpeg.cpp); PNG fixtures through Pillow."""
    if path.lower().endswith((".jpg", ".jpeg")):
        import ctypes as C
        from . import _lib
        lib = _lib.load().lib
        lib.rt_decode_jpeg_rgb8.restype = C.c_void_p
        lib.rt_decode_jpeg_rgb8.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.rt_free.argtypes = [C.c_void_p]
        data = open(path, "rb").read()
        w, h = C.c_uint32(), C.c_uint32()
        ptr = lib.rt_decode_jpeg_rgb8(data, len(data), C.byref(w), C.byref(h))
        if not ptr:
            raise RuntimeError(lib.rt_last_error().decode())
        out = C.string_at(ptr, 3 * w.value * h.value)
        lib.rt_free(ptr)
        return out, w.value, h.value
    from PIL import Image
    im = Image.open(path).convert("RGB")
    return im.tobytes(), im.size[0], im.size[1]


def load_earthmap():
    """The reference's earth texture (src/main.rs:248,491-495): assets/earthmap.jpg is a byte copy of its 1024x512 baseline
    4:4:4 asset, decoded by the library's own JPEG ingest (csrc/rt_jpeg.cpp) -> (rgb8 bytes, 1024, 512)."""
    return load_image_rgb8(asset_path("earthmap.jpg"))


def asset_path(name: str) -> str:
    """Data assets shipped with the package (raytracinginrust_amd/assets/: byte copies of the reference's earthmap.jpg and teapot.obj,
    which cannot be read from /root/reference at run time)."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", name)
