"""oracle/orc.py — ctypes wrapper of the CPU oracle (oracle/oracle.cpp).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product
package.  Exposes the oracle's builder API through the same `Backend`/`SceneBuilder` classes the product
uses (prefix `orc_`), plus `render`, event counters and function-level entry points for KATs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from raytracinginrust_amd.api import Backend, CameraParams, c_double_p  # noqa: E402  (builder API shape only)

LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
COUNTER_NAMES = ("samples", "world_hits", "bvh_nodes", "rect_tests", "sphere_tests", "msphere_tests", "tri_tests",
                 "xforms", "medium_tests", "shades", "light_pdf", "light_random", "texels", "perlin_evals",
                 "nonfinite", "bounces")

# Algorithmic record sizes (bytes) of SURVEY.md §8(d): f32-compact records, independent of the device layout.
RECORD_BYTES = {"bvh_nodes": 32, "rect_tests": 24, "sphere_tests": 20, "msphere_tests": 32, "tri_tests": 40,
                "xforms": 12, "medium_tests": 12, "shades": 16, "light_pdf": 24, "light_random": 24,
                "texels": 3, "perlin_evals": 120}
FRAMEBUFFER_BYTES_PER_PIXEL = 12

_backend = None
NOCOUNT_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_nocount.so")
BUILD_INFO = {"compiler": None,      # filled in by load_nocount() from the library's own __VERSION__ (orc_build_info)
              "flags": "-O3 -std=c++17 -ffp-contract=off -fno-fast-math -pthread, baseline x86-64 (no -march=native: built on one host, "
                       "run on another); event counters compiled out (-DORC_NO_COUNTERS)"}
_nocount = None


def load_nocount() -> Backend:
    """The oracle built with every event counter compiled out (bench.py's cpu_baseline leg): same arithmetic, no bookkeeping."""
    global _nocount
    if _nocount is None:
        if not os.path.exists(NOCOUNT_LIB_PATH):
            build()
        _nocount = _declare(C.CDLL(NOCOUNT_LIB_PATH))
        BUILD_INFO["compiler"] = _nocount.lib.orc_build_info().decode()
    return _nocount


OPCOUNT_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_opcount.so")
OP_KINDS = ("add", "mul", "div", "sqrt", "cmp", "minmax", "sin", "cos", "tan", "atan", "atan2", "acos", "log", "log2", "pow", "floor",
            "negabs", "cvt")
_opcount = None


def load_opcount() -> Backend:
    """The oracle built with `double` replaced by a counting stand-in (oracle/orc_opcount.h, -DORC_COUNT_OPS): same samples, and every
    f64 operation tallied by kind — `op_counts` reads the tallies."""
    global _opcount
    if _opcount is None:
        if not os.path.exists(OPCOUNT_LIB_PATH):
            build()
        _opcount = _declare(C.CDLL(OPCOUNT_LIB_PATH))
        assert _opcount.lib.orc_counts_ops() == 1 and _opcount.lib.orc_op_kinds() == len(OP_KINDS)
    return _opcount


def op_counts(be: Backend) -> dict:
    """f64 operations by kind executed by the renders of `be`'s library since the last call (reset on read)."""
    n = np.zeros(len(OP_KINDS), dtype=np.uint64)
    be.lib.orc_op_counts(C.c_void_p(n.ctypes.data))
    return dict(zip(OP_KINDS, [int(x) for x in n]))


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load() -> Backend:
    global _backend
    if _backend is not None:
        return _backend
    path = os.environ.get("ORC_LIB", LIB_PATH)          # ORC_LIB: the sanitizer build (tests/test_sanitizers.py)
    if not os.path.exists(path):
        build()
    _backend = _declare(C.CDLL(path))
    return _backend


def _declare(lib) -> Backend:
    be = Backend(lib, "orc_")
    cam_p = C.POINTER(CameraParams)
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [C.c_void_p, cam_p, c_double_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                               C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_hardware_threads.restype = C.c_int
    lib.orc_build_info.restype = C.c_char_p
    lib.orc_cube_hit_batch.restype = None
    lib.orc_cube_hit_batch.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_room_hit_batch.restype = None
    lib.orc_room_hit_batch.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    d3 = c_double_p
    lib.orc_sphere_uv.argtypes = [d3, d3]
    lib.orc_reflect.argtypes = [d3, d3, d3]
    lib.orc_refract.argtypes = [d3, d3, C.c_double, d3]
    lib.orc_reflectance.restype = C.c_double
    lib.orc_reflectance.argtypes = [C.c_double, C.c_double]
    lib.orc_onb.argtypes = [d3, d3]
    lib.orc_aabb_hit.restype = C.c_int
    lib.orc_aabb_hit.argtypes = [d3, d3, d3, d3, C.c_double, C.c_double]
    lib.orc_hit.restype = C.c_int
    lib.orc_hit.argtypes = [C.c_void_p, C.c_int, d3, d3, C.c_double, C.c_double, C.c_double, C.c_void_p, d3]
    lib.orc_pdf_value.restype = C.c_double
    lib.orc_pdf_value.argtypes = [C.c_void_p, C.c_int, d3, d3]
    lib.orc_random.argtypes = [C.c_void_p, C.c_int, d3, C.c_void_p, d3]
    lib.orc_lights_pdf_value.restype = C.c_double
    lib.orc_lights_pdf_value.argtypes = [C.c_void_p, d3, d3]
    lib.orc_cosine_generate.argtypes = [d3, C.c_void_p, d3]
    lib.orc_cosine_value.restype = C.c_double
    lib.orc_cosine_value.argtypes = [d3, d3]
    lib.orc_texture_value.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, d3, d3]
    lib.orc_bounding_box.restype = C.c_int
    lib.orc_bounding_box.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, d3]
    lib.orc_camera_ray.argtypes = [cam_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, d3]
    lib.orc_brdf.argtypes = [C.c_void_p, C.c_int, d3, d3, d3, d3]
    lib.orc_brdf_pdf_value.restype = C.c_double
    lib.orc_brdf_pdf_value.argtypes = [C.c_void_p, C.c_int, d3, d3, d3]
    lib.orc_brdf_pdf_generate.argtypes = [C.c_void_p, C.c_int, d3, d3, C.c_void_p, d3]
    lib.orc_ray_color.argtypes = [C.c_void_p, d3, d3, C.c_double, d3, C.c_uint64, C.c_void_p, d3]
    return be


def _d(*xs):
    return (C.c_double * len(xs))(*[float(x) for x in xs])


def hardware_threads() -> int:
    return int(load().lib.orc_hardware_threads())


def render(b, cam, background, W, H, spp, max_depth, seed=0x5EED, want_samples=False, want_counters=False,
           nthreads=0, mode=0, rows=None):
    """Per-pixel sums (H, W, 3) f64 in output order; optionally (H, W, spp, 3) samples and the event counters.  The scene `b`
    decides which build of the oracle runs (the counting one from load(), or load_nocount())."""
    be = b.b
    out = np.zeros((H, W, 3), dtype=np.float64)
    samples = np.zeros((H, W, spp, 3), dtype=np.float64) if want_samples else None
    cnt = np.zeros(len(COUNTER_NAMES), dtype=np.uint64)
    r0, r1 = (0, H) if rows is None else rows
    rc = be.lib.orc_render(b.h, C.byref(cam), _d(*background), W, H, spp, max_depth, seed, r0, r1, nthreads, mode,
                           out.ctypes.data, samples.ctypes.data if want_samples else None, cnt.ctypes.data)
    if rc != 0:
        raise RuntimeError(be.fn("scene_error")(b.h).decode())
    res = [out]
    if want_samples:
        res.append(samples)
    if want_counters:
        res.append(dict(zip(COUNTER_NAMES, [int(x) for x in cnt])))
    return res[0] if len(res) == 1 else tuple(res)


def set_isotropic_scatters(b, on: bool):
    """Opt-in, non-reference: Isotropic runs its old `scatter` (mat.rs:417-421) — mirrors the product's RT_ISOTROPIC_SCATTER."""
    load().lib.orc_scene_set_isotropic_scatters(C.c_void_p(b.h), 1 if on else 0)


def algorithmic_bytes_per_sample(counters: dict, spp: int) -> float:
    """sum over path events x record size, plus the framebuffer write amortised over spp."""
    n = counters["samples"]
    total = sum(counters[k] * v for k, v in RECORD_BYTES.items())
    return total / n + FRAMEBUFFER_BYTES_PER_PIXEL / spp


def timed_render(b, cam, background, W, H, spp, max_depth, seed=0x5EED, nthreads=0, mode=0, rows=None):
    t = time.perf_counter()
    out = render(b, cam, background, W, H, spp, max_depth, seed, nthreads=nthreads, mode=mode, rows=rows)
    return out, time.perf_counter() - t


# ---- function-level entry points (KATs)
def sphere_uv(p):
    be = load(); uv = _d(0, 0); be.lib.orc_sphere_uv(_d(*p), uv); return uv[0], uv[1]


def hit(b, h, o, d, time_=0.0, t_min=0.00001, t_max=float("inf"), rng=None):
    be = load(); out = (C.c_double * 10)()
    ok = be.lib.orc_hit(b.h, h.id, _d(*o), _d(*d), time_, t_min, t_max, rng.h if rng else None, out)
    if not ok:
        return None
    v = list(out)
    return {"position": v[0:3], "normal": v[3:6], "t": v[6], "u": v[7], "v": v[8], "front_face": bool(v[9])}


def pdf_value(b, h, o, v):
    return load().lib.orc_pdf_value(b.h, h.id, _d(*o), _d(*v))


def random(b, h, o, rng):
    out = _d(0, 0, 0); load().lib.orc_random(b.h, h.id, _d(*o), rng.h, out); return list(out)


def ray_color(b, o, d, time_, background, depth, rng):
    out = _d(0, 0, 0); load().lib.orc_ray_color(b.h, _d(*o), _d(*d), time_, _d(*background), depth, rng.h, out); return list(out)


def brdf(b, mat, r_in, r_out, normal):
    out = _d(0, 0, 0); load().lib.orc_brdf(b.h, mat.id, _d(*r_in), _d(*r_out), _d(*normal), out); return list(out)


def brdf_pdf_value(b, mat, r_in, r_out, normal):
    return load().lib.orc_brdf_pdf_value(b.h, mat.id, _d(*r_in), _d(*r_out), _d(*normal))


def brdf_pdf_generate(b, mat, r_in, normal, rng):
    out = _d(0, 0, 0); load().lib.orc_brdf_pdf_generate(b.h, mat.id, _d(*r_in), _d(*normal), rng.h, out); return list(out)
