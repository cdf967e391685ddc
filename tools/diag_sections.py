"""Section shares of the lock-step kernel from a -DRT_DIAG build (tools/mkab.sh diag ... -DRT_DIAG); shares only, never a timing.
usage: RT_WORKLOADS=C2,C3,C1 python tools/diag_sections.py [spp]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/diag.so')
import torch
from PIL import Image
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = ['refill', 'flush+init', 'world_hit', 'finalize', 'shade', 'terminate']
for key in os.environ.get('RT_WORKLOADS', 'C2').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    R.render(b, cam, bg, w.W, w.H, min(spp, w.spp), w.max_depth, flags=R.RT_LOCKSTEP_BVH)
    cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
    st = R.last_stats(b); tot = sum(cyc[:6])
    print(f'{key} ({w.scene}), {st["wave_iterations"]} wave iterations, alive lanes {st["live_lane_iterations"] / (64 * st["wave_iterations"]):.3f}:')
    if cyc[6]: print(f'    rect tests per wavefront: {cyc[6]}, with no lane in [t_min, closest]: {cyc[7]} ({cyc[7] / cyc[6] * 100:.1f} %)')
    for n, c in zip(names, cyc):
        print(f'    {n:12s} {c / tot * 100:6.2f} %   {c / st["wave_iterations"]:10.0f} wave-cycles per iteration')
