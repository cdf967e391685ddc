"""Python mirror of the reference's scene-builder API over a C library.

The reference (4meame/RayTracingInRust) builds scenes with Rust constructors
(`Sphere::new`, `AARect::new`, `Translate::new(Rotate::new(Axis::Y, Cube::new(..), -18.0), ..)`,
src/main.rs:278-311).  `SceneBuilder` exposes the same names, one method per constructor, and
forwards each call immediately (construction order matters: `NoiseTexture::new` draws from the
random stream, src/perlin.rs:67-75) to a C library that exports the builder entry points of
include/rt_amd.h under a prefix.  The product library (`librt_amd.so`, prefix ``rt_``) is the
default; tests pass the CPU oracle's library (prefix ``orc_``), which exports the same builder
names, so one scene function feeds both sides of a parity check.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Sequence

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)
c_int_p = C.POINTER(C.c_int)


class Plane:      # src/rect.rs:9-13
    XY, XZ, YZ = 0, 1, 2


class Axis:       # src/rotate.rs:8-12
    X, Y, Z = 0, 1, 2


def _v3(v: Sequence[float]):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class CameraParams(C.Structure):
    """Arguments of Camera::new (src/camera.rs:19)."""
    _fields_ = [("lookfrom", C.c_double * 3), ("lookat", C.c_double * 3), ("vup", C.c_double * 3),
                ("vfov", C.c_double), ("aspect", C.c_double), ("aperture", C.c_double),
                ("focus_dist", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


def Camera(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time0, time1) -> CameraParams:
    c = CameraParams()
    c.lookfrom[:] = [float(x) for x in lookfrom]
    c.lookat[:] = [float(x) for x in lookat]
    c.vup[:] = [float(x) for x in vup]
    c.vfov, c.aspect, c.aperture = float(vfov), float(aspect_ratio), float(aperture)
    c.focus_dist, c.time0, c.time1 = float(focus_dist), float(time0), float(time1)
    return c


class Backend:
    """A loaded C library plus the symbol prefix of its builder API."""

    _BUILDER_SIGS = {
        "scene_create": (C.c_void_p, []),
        "scene_destroy": (None, [C.c_void_p]),
        "scene_error": (C.c_char_p, [C.c_void_p]),
        "rng_create": (C.c_void_p, [C.c_uint64, C.c_uint32]),
        "rng_destroy": (None, [C.c_void_p]),
        "rng_f64": (C.c_double, [C.c_void_p]),
        "rng_range": (C.c_double, [C.c_void_p, C.c_double, C.c_double]),
        "rng_bool": (C.c_int, [C.c_void_p]),
        "rng_index": (C.c_uint32, [C.c_void_p, C.c_uint32]),
        "rng_u32": (C.c_uint32, [C.c_void_p]),
        "rng_path": (None, [C.c_uint64, C.c_uint32, C.c_uint32, c_u32_p]),
        "texture_constant": (C.c_int, [C.c_void_p, c_double_p]),
        "texture_check": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "texture_noise": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
        "texture_image": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]),
        "material_lambertian": (C.c_int, [C.c_void_p, C.c_int]),
        "material_metal": (C.c_int, [C.c_void_p, c_double_p, C.c_double]),
        "material_dielectric": (C.c_int, [C.c_void_p, C.c_double]),
        "material_diffuse_light": (C.c_int, [C.c_void_p, C.c_int]),
        "material_isotropic": (C.c_int, [C.c_void_p, C.c_int]),
        "material_pbr": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
        "sphere": (C.c_int, [C.c_void_p, c_double_p, C.c_double, C.c_int]),
        "moving_sphere": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int]),
        "aarect": (C.c_int, [C.c_void_p, C.c_int] + [C.c_double] * 5 + [C.c_int]),
        "cube": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int]),
        "triangle": (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
        "list_create": (C.c_int, [C.c_void_p]),
        "list_push": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "mesh": (C.c_int, [C.c_void_p, c_double_p, C.c_uint32, c_u32_p, C.c_uint32, C.c_int]),
        "flip_normal": (C.c_int, [C.c_void_p, C.c_int]),
        "translate": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
        "rotate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double]),
        "constant_medium": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int]),
        "bvh": (C.c_int, [C.c_void_p, c_int_p, C.c_uint32, C.c_double, C.c_double]),
        "bvh_of_list": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double]),
        "scene_set_world": (C.c_int, [C.c_void_p, C.c_int]),
        "lights_push": (C.c_int, [C.c_void_p, C.c_int]),
        "camera_fields": (None, [C.POINTER(CameraParams), c_double_p]),
        "format_color": (None, [c_double_p, C.c_uint64, c_u64_p]),
    }

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib, self.prefix = lib, prefix
        for name, (res, args) in self._BUILDER_SIGS.items():
            fn = getattr(lib, prefix + name)
            fn.restype, fn.argtypes = res, args

    def fn(self, name: str):
        return getattr(self.lib, self.prefix + name)


class Rng:
    """Seeded stand-in for rand::thread_rng() on the host side (scene construction)."""

    def __init__(self, backend: Backend, seed: int, stream: int):
        self.b = backend
        self.h = backend.fn("rng_create")(seed, stream)

    def __del__(self):
        try:
            self.b.fn("rng_destroy")(self.h)
        except Exception:
            pass

    def gen_f64(self) -> float:                 # rng.gen::<f64>()
        return self.b.fn("rng_f64")(self.h)

    def gen_range(self, a: float, b: float) -> float:   # rng.gen_range(a..b)
        return self.b.fn("rng_range")(self.h, a, b)

    def gen_bool(self) -> bool:
        return bool(self.b.fn("rng_bool")(self.h))

    def gen_index(self, n: int) -> int:
        return self.b.fn("rng_index")(self.h, n)

    def color_random(self, a: float, b: float):   # Color::random(a..b), src/vec.rs:70-76
        return (self.gen_range(a, b), self.gen_range(a, b), self.gen_range(a, b))


@dataclass(frozen=True)
class Handle:
    kind: str   # "texture" | "material" | "hittable"
    id: int


class SceneError(RuntimeError):
    pass


class SceneBuilder:
    """One C scene; methods are named after the reference constructors they stand for."""

    def __init__(self, backend: Backend):
        self.b = backend
        self.h = backend.fn("scene_create")()
        if not self.h:
            raise SceneError("scene_create failed")

    def close(self):
        if self.h:
            self.b.fn("scene_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, kind: str, rc: int) -> Handle:
        if rc < 0:
            raise SceneError(self.b.fn("scene_error")(self.h).decode())
        return Handle(kind, rc)

    def _call(self, name, *args):
        return self.b.fn(name)(self.h, *args)

    # textures (src/texture.rs)
    def ConstantTexture(self, color) -> Handle:
        return self._chk("texture", self._call("texture_constant", _v3(color)))

    def CheckTexture(self, odd: Handle, even: Handle) -> Handle:
        return self._chk("texture", self._call("texture_check", odd.id, even.id))

    def NoiseTexture(self, scale: float, rng: Rng) -> Handle:
        return self._chk("texture", self._call("texture_noise", float(scale), rng.h))

    def ImageTexture(self, data: bytes, width: int, height: int) -> Handle:
        assert len(data) == 3 * width * height
        return self._chk("texture", self._call("texture_image", bytes(data), width, height))

    # materials (src/mat.rs)
    def Lambertian(self, albedo: Handle) -> Handle:
        return self._chk("material", self._call("material_lambertian", albedo.id))

    def Metal(self, albedo, fuzz: float) -> Handle:
        return self._chk("material", self._call("material_metal", _v3(albedo), float(fuzz)))

    def Dielectric(self, ir: float) -> Handle:
        return self._chk("material", self._call("material_dielectric", float(ir)))

    def DiffuseLight(self, emit: Handle) -> Handle:
        return self._chk("material", self._call("material_diffuse_light", emit.id))

    def Isotropic(self, albedo: Handle) -> Handle:
        return self._chk("material", self._call("material_isotropic", albedo.id))

    def PBR(self, base_color: Handle, metallic, subsurface, specular, roughness, specular_tint, anisotropic, sheen,
            sheen_tint, clearcoat, clearcoat_gloss) -> Handle:
        """PBR::new (src/mat.rs:101), the principled material; same argument order."""
        p = (C.c_double * 10)(*[float(x) for x in (metallic, subsurface, specular, roughness, specular_tint, anisotropic,
                                                   sheen, sheen_tint, clearcoat, clearcoat_gloss)])
        return self._chk("material", self._call("material_pbr", base_color.id, p))

    # hittables
    def Sphere(self, center, radius, material: Handle) -> Handle:
        return self._chk("hittable", self._call("sphere", _v3(center), float(radius), material.id))

    def MovingSphere(self, c0, c1, t0, t1, radius, material: Handle) -> Handle:
        return self._chk("hittable", self._call("moving_sphere", _v3(c0), _v3(c1), float(t0), float(t1), float(radius), material.id))

    def AARect(self, plane: int, a0, a1, b0, b1, k, material: Handle) -> Handle:
        return self._chk("hittable", self._call("aarect", plane, float(a0), float(a1), float(b0), float(b1), float(k), material.id))

    def Cube(self, mn, mx, material: Handle) -> Handle:
        return self._chk("hittable", self._call("cube", _v3(mn), _v3(mx), material.id))

    def Triangle(self, vertices, material: Handle) -> Handle:
        flat = [float(x) for v in vertices for x in v]
        return self._chk("hittable", self._call("triangle", (C.c_double * 9)(*flat), material.id))

    def HittableList(self) -> "ListHandle":
        return ListHandle(self, self._chk("hittable", self._call("list_create")).id)

    def Mesh(self, positions, indices, material: Handle) -> "ListHandle":
        """Mesh::new (src/mesh.rs:16-31); returns the `tris` list."""
        flat = [float(x) for p in positions for x in p]
        pos = (C.c_double * len(flat))(*flat)
        idx = (C.c_uint32 * len(indices))(*[int(i) for i in indices])
        h = self._chk("hittable", self._call("mesh", pos, len(flat) // 3, idx, len(indices), material.id))
        return ListHandle(self, h.id)

    def FlipNormal(self, h: Handle) -> Handle:
        return self._chk("hittable", self._call("flip_normal", h.id))

    def Translate(self, h: Handle, offset) -> Handle:
        return self._chk("hittable", self._call("translate", h.id, _v3(offset)))

    def Rotate(self, axis: int, h: Handle, angle: float) -> Handle:
        return self._chk("hittable", self._call("rotate", axis, h.id, float(angle)))

    def ConstantMedium(self, boundary: Handle, density: float, texture: Handle) -> Handle:
        return self._chk("hittable", self._call("constant_medium", boundary.id, float(density), texture.id))

    def BVH(self, hittables, time0: float, time1: float) -> Handle:
        """BVH::new(Vec<Box<dyn Hittable>>, t0, t1) — accepts a python list of handles or a ListHandle (`obj.tris.list`)."""
        if isinstance(hittables, ListHandle):
            return self._chk("hittable", self._call("bvh_of_list", hittables.id, float(time0), float(time1)))
        ids = (C.c_int * len(hittables))(*[h.id for h in hittables])
        return self._chk("hittable", self._call("bvh", ids, len(hittables), float(time0), float(time1)))

    def set_scene(self, world: Handle, lights: Sequence[Handle]):
        """The `(world, lights)` pair a scene fn returns (src/main.rs:153)."""
        if self._call("scene_set_world", world.id) < 0:
            raise SceneError(self.b.fn("scene_error")(self.h).decode())
        for l in lights:
            if self._call("lights_push", l.id) < 0:
                raise SceneError(self.b.fn("scene_error")(self.h).decode())


class ListHandle(Handle):
    def __init__(self, builder: SceneBuilder, id: int):
        object.__setattr__(self, "kind", "hittable")
        object.__setattr__(self, "id", id)
        object.__setattr__(self, "_b", builder)

    def push(self, h: Handle):
        if self._b._call("list_push", self.id, h.id) < 0:
            raise SceneError(self._b.b.fn("scene_error")(self._b.h).decode())


def camera_fields(backend: Backend, cam: CameraParams):
    out = (C.c_double * 21)()
    backend.fn("camera_fields")(C.byref(cam), out)
    return list(out)


def format_color(backend: Backend, rgb_sum, spp: int):
    out = (C.c_uint64 * 3)()
    backend.fn("format_color")(_v3(rgb_sum), spp, out)
    return tuple(int(x) for x in out)
