"""Traversal-pass vs advance-pass shares of the persistent-traversal kernel from a -DRT_DIAG build (tools/mkab.sh diag ... -DRT_DIAG):
traversal passes, and inside the advance passes: shade (hit record + material), new paths (queue pop / refill), accumulator flush +
path init, list walk up to the next BVH object."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/diag.so')
import torch
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
flags = int(os.environ.get('RT_FLAGS', '0'))
for key in os.environ.get('RT_WORKLOADS', 'C4').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    R.render(b, cam, bg, w.W, w.H, int(sys.argv[1]) if len(sys.argv) > 1 else 32, w.max_depth, flags=flags)
    cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
    tv = R.last_traversal_stats(b); tot = sum(cyc[:6])
    lf = [tv['leaf_steps'], tv['leaf_lanes']]
    nb, nl = tv['traversal_steps'] - lf[0], lf[0]
    if nl:
        print(f"{key}: box steps {nb/1e6:.1f}M at {cyc[0]/max(1,nb):.0f} wave-cycles, util {(tv['traversal_lanes']-lf[1])/max(1,64*nb):.2f} ({cyc[0]/tot*100:.1f} %); "
              f"leaf steps {nl/1e6:.1f}M at {cyc[5]/max(1,nl):.0f} wave-cycles, util {lf[1]/max(1,64*nl):.2f} ({cyc[5]/tot*100:.1f} %)")
    print(f"{key}: {R.last_kernel_ms(b):.1f} ms; traversal {(cyc[0]+cyc[5])/tot*100:5.1f} % ({(cyc[0]+cyc[5])/max(1,tv['traversal_steps']):7.0f} wave-cycles/step, util {tv['traversal_lanes']/max(1,64*tv['traversal_steps']):.2f}); "
          f"advance passes util {tv['advance_lanes']/max(1,64*tv['advance_passes']):.2f}, wave-cycles per pass: " +
          ", ".join(f"{n} {cyc[k]/max(1,tv['advance_passes']):.0f} ({cyc[k]/tot*100:.1f} %)" for k, n in ((1, 'shade'), (2, 'new paths'), (3, 'flush+init'), (4, 'list walk'))))
