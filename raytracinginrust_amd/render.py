"""Host-side mirror of the reference's render loop (src/main.rs:767-835) over the C-ABI.

`render()` is the single call that replaces the `for j / for i / into_par_iter().map().sum()` nest;
`render_tiles_device()` is the tile-sharded form used one-process-per-GPU (see dist.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .api import CameraParams, SceneBuilder

RT_F64, RT_F32, RT_STOP_ON_ZERO, RT_ISOTROPIC_SCATTER, RT_NEAR_FIRST_BVH = 0, 1, 2, 4, 8
RT_PERSISTENT_BVH, RT_LOCKSTEP_BVH = 16, 32
RT_MULTI_COLLECTIVE = 64
RT_SPECULATE_BVH, RT_NO_SPECULATE_BVH = 1024, 2048
FLATTEN_COUNT_NAMES = ("objects", "ops", "rects", "spheres", "moving_spheres", "triangles", "bvh_nodes",
                       "materials", "textures", "lights", "media", "perlins")


class RenderError(RuntimeError):
    pass


def _err(be) -> str:
    return be.lib.rt_last_error().decode()


def device_count() -> int:
    return _lib.load().lib.rt_device_count()


def flatten(b: SceneBuilder) -> dict:
    """Flatten the Hittable tree into the device scene (host only) and return the table sizes."""
    be = _lib.load()
    counts = (C.c_uint32 * 12)()
    if be.lib.rt_scene_flatten(b.h, counts) != 0:
        raise RenderError(_err(be))
    return dict(zip(FLATTEN_COUNT_NAMES, [int(x) for x in counts]))


OBJECT_FIELDS = ("geom_kind", "geom_first", "geom_count", "first_op", "n_ops", "medium", "is_cube", "nest")


def debug_objects(b: SceneBuilder, top_only: bool = True) -> list:
    """The flattened object table (rt_debug_objects; host only): one dict per object, the world's top-level objects in the order the
    kernels search them.  is_cube: 1 = a Cube's six faces; 2 | map << 8 = a ROOM (bare AARects that are faces of one box, tested through
    the Cube fast path: rt_flatten.cpp form_room) — map: three bits per face in cube.rs:17-24 order, the wall's place in the room's run
    of rect records or 7 = no such wall; first_op then holds, five bits per wall, the tie-rule index."""
    be = _lib.load()
    be.lib.rt_debug_objects.restype = C.c_int
    be.lib.rt_debug_objects.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    n_top = C.c_uint32(0)
    n = be.lib.rt_debug_objects(b.h, None, 0, C.byref(n_top))
    if n < 0:
        raise RenderError(_err(be))
    out = np.zeros((max(n, 1), 8), np.uint32)
    if be.lib.rt_debug_objects(b.h, out.ctypes.data, n, C.byref(n_top)) != n:
        raise RenderError(_err(be))
    rows = out[: (int(n_top.value) if top_only else n)]
    return [dict(zip(OBJECT_FIELDS, [int(x) for x in r])) for r in rows]


RT_BVH_MEDIAN, RT_BVH_SAH = 0, 1


def set_bvh_builder(b: SceneBuilder, mode: int) -> None:
    """RT_BVH_MEDIAN: the reference's BVH::new (default); RT_BVH_SAH: opt-in binned surface-area heuristic."""
    be = _lib.load()
    if be.lib.rt_scene_set_bvh_builder(b.h, mode) != 0:
        raise RenderError(_err(be))


def set_traversal_schedule(b: SceneBuilder, start_at: int = 40, stop_below: int = 24, leaf_share64: int = 16) -> None:
    """Tuning knob of the persistent-traversal loop (scheduling only)."""
    be = _lib.load()
    if be.lib.rt_scene_set_traversal_schedule(b.h, start_at, stop_below, leaf_share64) != 0:
        raise RenderError(_err(be))


def prepare(b: SceneBuilder, flags: int = RT_F64) -> None:
    """Flatten, upload and load the kernel now instead of inside the first render (no launch)."""
    be = _lib.load()
    if be.lib.rt_scene_prepare(b.h, flags) != 0:
        raise RenderError(_err(be))


LOOP_SHAPES = ("list", "lock-step", "persistent")
LOOP_CHOSEN_BY = ("the scene leaves no choice", "size rule", "calibration of this view", "caller's flag", "rt_scene_set_loop_shape")


def calibrate(b: SceneBuilder, cam: CameraParams, background, W: int, H: int, spp: int, max_depth: int,
              seed: int = 0x5EED, flags: int = RT_F64) -> None:
    """rt_scene_calibrate: mesh scenes measure now, synchronously, which loop shape is faster for this view (four small launches);
    a no-op for every other scene.  The asynchronous entry points (render_tiles_device, render_multi_device) never do it themselves."""
    be = _lib.load()
    bg = (C.c_double * 3)(*[float(x) for x in background])
    if be.lib.rt_scene_calibrate(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags) != 0:
        raise RenderError(_err(be))


def set_loop_shape(b: SceneBuilder, shape: int) -> None:
    """rt_scene_set_loop_shape: 1 persistent traversal, 0 lock-step (every view, until the scene changes), -1 forget."""
    be = _lib.load()
    if be.lib.rt_scene_set_loop_shape(b.h, shape) != 0:
        raise RenderError(_err(be))


def stored_loop_shape(b: SceneBuilder) -> int:
    """rt_scene_loop_shape: 1 persistent traversal, 0 lock-step, -1 nothing stored."""
    return int(_lib.load().lib.rt_scene_loop_shape(b.h))


def last_loop_info(b: SceneBuilder) -> dict:
    """rt_last_loop_info: the loop shape and instantiation the most recent launch ran, how the shape was chosen, and the stored
    calibration's kernel times."""
    be = _lib.load()
    out = (C.c_int32 * 4)(); ms = (C.c_float * 2)()
    if be.lib.rt_last_loop_info(b.h, out, ms) != 0:
        raise RenderError(_err(be))
    t = "float" if out[3] else "double"
    return {"shape": LOOP_SHAPES[out[0]], "feats": int(out[1]), "kernel": f"rt::pathtrace_kernel<{t}, {int(out[1])}u>",
            "chosen_by": LOOP_CHOSEN_BY[out[2]],
            "calibration_ms": {"lock-step": float(ms[0]), "persistent": float(ms[1])} if ms[0] > 0 or ms[1] > 0 else None}


def render(b: SceneBuilder, cam: CameraParams, background, W: int, H: int, spp: int, max_depth: int,
           seed: int = 0x5EED, flags: int = RT_F64, want_samples: bool = False):
    """Per-pixel sums of ray_color over `spp` samples, shape (H, W, 3) f64, row 0 = top (what `.sum()`
    yields at src/main.rs:830, in the order the reference prints pixels).  With want_samples also returns
    the (H, W, spp, 3) per-sample radiance."""
    be = _lib.load()
    out = np.zeros((H, W, 3), dtype=np.float64)
    bg = (C.c_double * 3)(*[float(x) for x in background])
    if want_samples:
        samples = np.zeros((H, W, spp, 3), dtype=np.float64)
        rc = be.lib.rt_render_samples(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags, out.ctypes.data, samples.ctypes.data)
    else:
        samples = None
        rc = be.lib.rt_render(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags, out.ctypes.data)
    if rc != 0:
        raise RenderError(_err(be))
    return (out, samples) if want_samples else out


def render_multi(b: SceneBuilder, cam: CameraParams, background, W: int, H: int, spp: int, max_depth: int, device_mask: int = 0,
                 seed: int = 0x5EED, flags: int = RT_F64, tile_px: int = 0):
    """The whole frame on the GPUs of this node selected by `device_mask` (bit d = HIP device d; 0 = all visible) from ONE call:
    per-device replicas of the scene, tiles dealt round-robin, one RCCL gather to the first device, un-permuted there
    (rt_render_multi, csrc/rt_multi.cpp).  Returns the (H, W, 3) per-pixel sums like render()."""
    be = _lib.load()
    out = np.zeros((H, W, 3), dtype=np.float64)
    bg = (C.c_double * 3)(*[float(x) for x in background])
    if be.lib.rt_render_multi(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags, device_mask, tile_px, out.ctypes.data) != 0:
        raise RenderError(_err(be))
    return out


def render_multi_device(b: SceneBuilder, cam: CameraParams, background, W: int, H: int, spp: int, max_depth: int, device_mask: int = 0,
                        seed: int = 0x5EED, flags: int = RT_F64, tile_px: int = 0) -> int:
    """rt_render_multi_device: enqueue the frame on the selected devices and return the DEVICE address (first selected device) of
    its W*H*3 per-pixel sums; `multi_sync` waits for it, `multi_frame` fetches it."""
    be = _lib.load()
    bg = (C.c_double * 3)(*[float(x) for x in background])
    ptr = C.c_void_p()
    if be.lib.rt_render_multi_device(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags, device_mask, tile_px, C.byref(ptr)) != 0:
        raise RenderError(_err(be))
    return int(ptr.value or 0)


def multi_sync(b: SceneBuilder) -> None:
    be = _lib.load()
    if be.lib.rt_multi_sync(b.h) != 0:
        raise RenderError(_err(be))


def multi_frame(b: SceneBuilder, W: int, H: int) -> np.ndarray:
    """The last rt_render_multi_device frame as an (H, W, 3) host array (waits for it)."""
    be = _lib.load()
    out = np.zeros((H, W, 3), dtype=np.float64)
    if be.lib.rt_multi_copy_frame(b.h, out.ctypes.data, out.size) != 0:
        raise RenderError(_err(be))
    return out


def last_multi_ms(b: SceneBuilder) -> dict:
    be = _lib.load()
    ms = (C.c_double * 4)()
    if be.lib.rt_last_multi_ms(b.h, ms) != 0:
        raise RenderError(_err(be))
    return {"slowest_kernel_ms": ms[0], "gather_ms": ms[1], "unpermute_ms": ms[2], "call_ms": ms[3]}


def last_multi_ranks(b: SceneBuilder) -> dict:
    """rt_last_multi_ranks: HIP device and kernel ms of every rank of the last rt_render_multi* frame, and the rank count RCCL
    reports for the communicator its gather ran on (0: no collective ran)."""
    be = _lib.load()
    n, coll = C.c_uint32(), C.c_uint32()
    dev = (C.c_int * 64)(); ms = (C.c_double * 64)()
    be.lib.rt_last_multi_ranks.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    if be.lib.rt_last_multi_ranks(b.h, 64, C.byref(n), dev, ms, C.byref(coll)) != 0:
        raise RenderError(_err(be))
    k = min(int(n.value), 64)
    return {"n_ranks": int(n.value), "devices": [int(dev[i]) for i in range(k)], "kernel_ms": [float(ms[i]) for i in range(k)],
            "collective_ranks": int(coll.value)}


def local_tiles(W: int, H: int, tile_px: int, rank: int, world: int) -> int:
    return int(_lib.load().lib.rt_local_tiles(W, H, tile_px, rank, world))


def render_tiles_device(b: SceneBuilder, cam: CameraParams, background, W: int, H: int, spp: int, max_depth: int,
                        seed: int, flags: int, tile_px: int, rank: int, world: int, d_out_ptr: int, d_out_bytes: int,
                        stream: int = 0) -> None:
    """Asynchronously render tiles t ≡ rank (mod world) into device memory at d_out_ptr on `stream`."""
    be = _lib.load()
    bg = (C.c_double * 3)(*[float(x) for x in background])
    rc = be.lib.rt_render_device(b.h, C.byref(cam), bg, W, H, spp, max_depth, seed, flags, tile_px, rank, world,
                                 C.c_void_p(d_out_ptr), d_out_bytes, C.c_void_p(stream))
    if rc != 0:
        raise RenderError(_err(be))


def kernel_time_total(b: SceneBuilder, reset: bool = False):
    """(total ms, launches) of this scene's kernels since the last reset; waits for launches in flight."""
    be = _lib.load()
    ms, n = C.c_double(), C.c_ulonglong()
    if be.lib.rt_kernel_time_total(b.h, C.byref(ms), C.byref(n), 1 if reset else 0) != 0:
        raise RenderError(_err(be))
    return float(ms.value), int(n.value)


def last_flush_count(b: SceneBuilder) -> int:
    """Accumulator flushes of the last launch (three f64 atomics to the frame each)."""
    be = _lib.load()
    out = C.c_ulonglong()
    if be.lib.rt_last_flush_count(b.h, C.byref(out)) != 0:
        raise RenderError(_err(be))
    return int(out.value)


def last_traversal_stats(b: SceneBuilder) -> dict:
    """BVH scenes: advance passes / traversal steps of the last launch and the lanes busy in each (summed over wavefronts)."""
    be = _lib.load()
    out = (C.c_ulonglong * 4)()
    if be.lib.rt_last_traversal_stats(b.h, out) != 0:
        raise RenderError(_err(be))
    leaf = (C.c_ulonglong * 2)()
    if be.lib.rt_last_leaf_steps(b.h, leaf) != 0:
        raise RenderError(_err(be))
    return {"advance_passes": out[0], "advance_lanes": out[1], "traversal_steps": out[2], "traversal_lanes": out[3],
            "leaf_steps": leaf[0], "leaf_lanes": leaf[1]}          # of the traversal steps: the primitive-test steps (the rest test boxes)


def last_launch_info(b: SceneBuilder) -> dict:
    """Geometry of the most recent launch (workgroups, threads, LDS bytes, BVH nodes staged in LDS / in the scene, workgroups per CU)."""
    be = _lib.load()
    out = (C.c_uint32 * 6)()
    be.lib.rt_last_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    if be.lib.rt_last_launch_info(b.h, out) != 0:
        raise RenderError(_err(be))
    return dict(zip(("workgroups", "threads", "lds_bytes", "bvh_nodes_in_lds", "bvh_nodes", "workgroups_per_cu"), [int(x) for x in out]))


def last_kernel_ms(b: SceneBuilder) -> float:
    be = _lib.load()
    ms = C.c_float()
    if be.lib.rt_last_kernel_ms(b.h, C.byref(ms)) != 0:
        raise RenderError(_err(be))
    return float(ms.value)


def last_stats(b: SceneBuilder) -> dict:
    be = _lib.load()
    st = (C.c_ulonglong * 3)()
    if be.lib.rt_last_stats(b.h, st) != 0:
        raise RenderError(_err(be))
    return {"nonfinite_samples": int(st[0]), "wave_iterations": int(st[1]), "live_lane_iterations": int(st[2])}


def format_image(rgb_sum: np.ndarray, spp: int) -> np.ndarray:
    """Vec3::format_color (src/vec.rs:125-131) over a whole frame -> (H, W, 3) uint8-range ints."""
    be = _lib.load()
    H, W, _ = rgb_sum.shape
    out = np.zeros((H, W, 3), dtype=np.uint64)
    flat = np.ascontiguousarray(rgb_sum, dtype=np.float64).reshape(-1, 3)
    o = out.reshape(-1, 3)
    fn = be.fn("format_color")
    buf = (C.c_uint64 * 3)()
    for p in range(flat.shape[0]):
        fn(flat[p].ctypes.data_as(C.POINTER(C.c_double)), spp, buf)
        o[p, 0], o[p, 1], o[p, 2] = buf[0], buf[1], buf[2]
    return out


def write_ppm(path: str, rgb_sum: np.ndarray, spp: int) -> None:
    """The reference's P3 emitter (src/main.rs:767-769,832)."""
    be = _lib.load()
    H, W, _ = rgb_sum.shape
    a = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    if be.lib.rt_write_ppm(path.encode(), a.ctypes.data, W, H, spp) != 0:
        raise RenderError(_err(be))
