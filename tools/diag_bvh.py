"""Traversal-pass vs advance-pass shares of the BVH kernel from a -DRT_DIAG build (tools/mkvariant.sh diag -DRT_DIAG)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/variants/diag.so')
import torch
from PIL import Image
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
for key in os.environ.get('RT_WORKLOADS', 'C1,C3,C4').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    R.render(b, cam, bg, w.W, w.H, 16, w.max_depth)
    cyc = (C.c_ulonglong * 6)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
    tv = R.last_traversal_stats(b); tot = cyc[0] + cyc[1]
    print(f"{key}: traversal {cyc[0]/tot*100:5.1f} %  ({cyc[0]/max(1,tv['traversal_steps']):8.0f} /step, util {tv['traversal_lanes']/max(1,64*tv['traversal_steps']):.2f})   advance {cyc[1]/tot*100:5.1f} %  ({cyc[1]/max(1,tv['advance_passes']):8.0f} /pass, util {tv['advance_lanes']/max(1,64*tv['advance_passes']):.2f})   {R.last_kernel_ms(b):.1f} ms")
