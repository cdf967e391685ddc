"""Section shares of the deferred-entry lock-step kernel from a -DRT_DIAG build (tools/mkab.sh diag "" -DRT_DIAG) and its parking
statistics; shares only, never a timing.   usage: RT_WORKLOADS=C3,C4 python tools/diag_defer.py [spp]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/ab/diag.so')
import torch
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
names = ['new paths + flush', 'list objects', 'root test + park', 'tree walks', 'take walked paths', 'hit record + material']
for key in os.environ.get('RT_WORKLOADS', 'C3').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    R.render(b, cam, bg, w.W, w.H, min(spp, w.spp), w.max_depth, flags=R.RT_DEFER_BVH)
    ms = R.last_kernel_ms(b)
    cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
    st = R.last_stats(b); tv = R.last_traversal_stats(b); tot = sum(cyc[:6])
    it = st["wave_iterations"]
    print(f'{key} ({w.scene}) {ms:.1f} ms: {it} wave iterations, {st["live_lane_iterations"] / (64 * it):.3f} of the lanes alive; '
          f'{tv["leaf_steps"] / it:.1f} paths parked per iteration, {tv["traversal_steps"]} walks of {tv["traversal_lanes"] / max(1, tv["traversal_steps"]):.1f} paths')
    for n, c in zip(names, cyc):
        print(f'    {n:22s} {c / tot * 100:6.2f} %   {c / it:10.0f} wave-cycles per iteration')
