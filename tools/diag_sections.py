"""Section shares of the kernel from a -DRT_DIAG build (tools only; never a timing source)."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, scenes
be = _lib.load_path(os.path.join(ROOT, 'raytracinginrust_amd/csrc/variants/diag.so'))
b, cam, bg = scenes.cornell_box(be)
W = H = 800; spp = 64
out = np.zeros((H, W, 3))
be.lib.rt_render(b.h, C.byref(cam), (C.c_double*3)(*bg), W, H, spp, 50, 0x5EED, 0, out.ctypes.data)
cyc = (C.c_ulonglong * 6)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
st = (C.c_ulonglong * 3)(); be.lib.rt_last_stats(b.h, st)
tot = sum(cyc); names = ['refill', 'flush+init', 'world_hit', 'finalize', 'shade', 'terminate']
for n, c in zip(names, cyc): print(f'{n:12s} {c/tot*100:6.2f} %   {c/st[1]:9.1f} cycles/iter')
print('iters', st[1], 'lane util', st[2]/(64*st[1]))
