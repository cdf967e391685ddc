"""Loader for the product library `csrc/librt_amd.so` (C-ABI of include/rt_amd.h).

Fails loudly when the library has not been built: there is no fallback renderer of any kind.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

from .api import Backend, CameraParams, c_double_p, c_u32_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_amd.so")

_backend = None


class LibraryMissing(RuntimeError):
    pass


def load() -> Backend:
    """Load librt_amd.so once and declare the render entry points."""
    global _backend
    if _backend is not None:
        return _backend
    _backend = load_path(os.environ.get("RT_AMD_LIB", LIB_PATH))     # RT_AMD_LIB: developer override (tools/)
    return _backend


def load_path(path: str) -> Backend:
    """Load a specific build of the library (tools/ab.py compares kernel variants in one process)."""
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} is missing: build it with `make -C raytracinginrust_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    # PyTorch first, when this process is going to use it: torch brings its own copy of the HIP runtime, and a process in which this
    # library's copy was loaded before it leaves torch without a device ("No HIP GPUs are available").  dist.py and bench.py need both,
    # so the preload is the default wherever torch is installed; RT_AMD_NO_TORCH_PRELOAD=1 skips it (a host that never imports torch
    # saves the import), and a torch that is present but broken must not make the renderer unloadable.
    if "torch" not in sys.modules and not os.environ.get("RT_AMD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:       # ImportError, OSError (a missing .so), RuntimeError (version mismatch): the renderer does not need torch
            pass
    lib = C.CDLL(path)
    be = Backend(lib, "rt_")
    cam_p = C.POINTER(CameraParams)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_device_count.restype = C.c_int
    lib.rt_scene_flatten.restype = C.c_int
    lib.rt_scene_flatten.argtypes = [C.c_void_p, c_u32_p]
    lib.rt_local_tiles.restype = C.c_uint32
    lib.rt_local_tiles.argtypes = [C.c_uint32] * 5
    common = [C.c_void_p, cam_p, c_double_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    lib.rt_render.restype = C.c_int
    lib.rt_render.argtypes = common + [C.c_void_p]
    lib.rt_render_samples.restype = C.c_int
    lib.rt_render_samples.argtypes = common + [C.c_void_p, C.c_void_p]
    lib.rt_render_device.restype = C.c_int
    lib.rt_render_device.argtypes = common + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.rt_render_multi.restype = C.c_int
    lib.rt_render_multi.argtypes = common + [C.c_uint32, C.c_uint32, C.c_void_p]
    if hasattr(lib, "rt_render_multi_device"):       # (tools/ab.py also loads builds of earlier rounds)
        lib.rt_render_multi_device.restype = C.c_int
        lib.rt_render_multi_device.argtypes = common + [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rt_multi_sync.restype = C.c_int
        lib.rt_multi_sync.argtypes = [C.c_void_p]
        lib.rt_multi_copy_frame.restype = C.c_int
        lib.rt_multi_copy_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.rt_last_multi_ms.restype = C.c_int
    lib.rt_last_multi_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.rt_kernel_time_total.restype = C.c_int
    lib.rt_kernel_time_total.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.c_int]
    lib.rt_last_flush_count.restype = C.c_int
    lib.rt_last_flush_count.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.rt_last_traversal_stats.restype = C.c_int
    lib.rt_last_traversal_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.rt_last_leaf_steps.restype = C.c_int
    lib.rt_last_leaf_steps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.rt_debug_bvh_links.restype = C.c_int
    lib.rt_debug_bvh_links.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rt_scene_set_traversal_schedule.restype = C.c_int
    lib.rt_scene_set_traversal_schedule.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.rt_scene_set_bvh_builder.restype = C.c_int
    lib.rt_scene_set_bvh_builder.argtypes = [C.c_void_p, C.c_int]
    lib.rt_scene_prepare.restype = C.c_int
    lib.rt_scene_prepare.argtypes = [C.c_void_p, C.c_uint32]
    if hasattr(lib, "rt_scene_calibrate"):           # (round 6; tools/ab.py also loads builds of earlier rounds)
        lib.rt_scene_calibrate.restype = C.c_int
        lib.rt_scene_calibrate.argtypes = common
        lib.rt_scene_set_loop_shape.restype = C.c_int
        lib.rt_scene_set_loop_shape.argtypes = [C.c_void_p, C.c_int]
        lib.rt_scene_loop_shape.restype = C.c_int
        lib.rt_scene_loop_shape.argtypes = [C.c_void_p]
        lib.rt_last_loop_info.restype = C.c_int
        lib.rt_last_loop_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    lib.rt_last_kernel_ms.restype = C.c_int
    lib.rt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.rt_last_stats.restype = C.c_int
    lib.rt_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.rt_write_ppm.restype = C.c_int
    lib.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    return be
