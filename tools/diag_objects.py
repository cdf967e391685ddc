"""Where world.hit's time goes by object class, from a -DRT_DIAG_OBJ build (tools/mkab.sh diagobj "" -DRT_DIAG_OBJ): wave-cycles in BVH objects
without wrappers / BVH objects behind wrappers / all other objects, and how many lanes' rays pass each BVH class's root box per call.
Lock-step loop only; shares and counts, never a timing.   usage: RT_WORKLOADS=C3,C4 python tools/diag_objects.py [spp]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/diagobj.so')
import torch
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for key in os.environ.get('RT_WORKLOADS', 'C3').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    R.render(b, cam, bg, w.W, w.H, min(spp, w.spp), w.max_depth, flags=R.RT_LOCKSTEP_BVH)
    ms = R.last_kernel_ms(b)
    cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
    leaf = (C.c_ulonglong * 2)(); be.lib.rt_last_leaf_steps(b.h, leaf)
    calls, lanes = cyc[5], leaf[0]
    tot = cyc[0] + cyc[1] + cyc[2]
    info = R.last_launch_info(b)
    wave_cycles_total = ms * 1e-3 * 2.4e9 * info["workgroups"] * info["threads"] / 64
    print(f'{key} ({w.scene}) {ms:.1f} ms: {calls} world.hit calls, {lanes / max(1, calls):.1f} lanes each; world.hit = {tot / wave_cycles_total * 100:.1f} % of the wave-cycles (at 2.4 GHz)')
    for n, c, l in (('BVH objects without wrappers', cyc[0], cyc[3]), ('BVH objects behind wrappers', cyc[1], cyc[4]), ('other objects', cyc[2], None)):
        print(f'    {n:30s} {c / max(1, tot) * 100:6.2f} % of world.hit, {c / max(1, calls):9.0f} wave-cycles per call' + (f', {l / max(1, calls):5.2f} lanes pass the root box per call' if l is not None else ''))
