import ctypes as C, sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracinginrust_amd import _lib, scenes, render as R
from raytracinginrust_amd.api import Backend, CameraParams
olib = C.CDLL('/root/repo/oracle/_build/liboracle.so')
obe = Backend(olib, 'orc_')
olib.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraParams), C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
def orender(b, cam, bg, W, H, spp, depth, seed=0x5EED, samples=False):
    out = np.zeros((H, W, 3)); smp = np.zeros((H,W,spp,3)) if samples else None
    rc = olib.orc_render(b.h, C.byref(cam), (C.c_double*3)(*bg), W, H, spp, depth, seed, 0, H, 0, 0, out.ctypes.data, smp.ctypes.data if samples else None, None)
    assert rc == 0
    return out, smp
be = _lib.load()
print('devices', R.device_count())
W=H=64; spp=32; depth=50
ob, ocam, obg = scenes.cornell_box(obe)
pb, pcam, pbg = scenes.cornell_box(be)
print(R.flatten(pb))
ref, rs = orender(ob, ocam, obg, W, H, spp, depth, samples=True)
t=time.time(); got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True); print('gpu time', time.time()-t, 'kernel ms', R.last_kernel_ms(pb), R.last_stats(pb))
d = np.abs(got-ref); print('pixel sum max abs diff', d.max(), 'max rel', (d/(np.abs(ref)+1e-300)).max(), 'ref mean', ref.mean())
ds = np.abs(gs-rs); nbad = (ds > 1e-9*(1+np.abs(rs))).any(axis=-1).sum(); print('samples: max abs diff', ds.max(), 'n samples differing >1e-9 rel:', nbad, 'of', W*H*spp, 'bit-identical frac', (gs==rs).all(axis=-1).mean())
# bigger timing run
W=H=800; spp=64
t=time.time(); got = R.render(pb, pcam, pbg, W, H, spp, depth); dt=time.time()-t
ms = R.last_kernel_ms(pb); st = R.last_stats(pb)
print('800x800x64: wall', dt, 'kernel ms', ms, 'Msamples/s', W*H*spp/ms/1e3, st, 'lane util', st['live_lane_iterations']/(64*st['wave_iterations']))
