"""Section shares of the exchange variant (-DRT_XCHG -DRT_DIAG build)."""
import ctypes as C, os, sys
ROOT = '/root/repo' if os.path.isdir('/root/repo') else os.getcwd(); sys.path.insert(0, ROOT)
os.environ['RT_AMD_LIB'] = os.path.join(ROOT, 'raytracinginrust_amd/csrc/abx/xdiag.so')
import torch
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
names = ['regen+flush+init', 'hit+record+key', 'barrier A', 'sort+write', 'barrier B', 'shade slots', 'barrier C', 'home read']
w = workloads.WORKLOADS['C2']
b, cam, bg = workloads.build(w, be, None)
R.render(b, cam, bg, w.W, w.H, 256, w.max_depth)
cyc = (C.c_ulonglong * 8)(); be.lib.rt_debug_section_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]; be.lib.rt_debug_section_cycles(b.h, cyc)
st = R.last_stats(b); tot = sum(cyc)
print(f'{st["wave_iterations"]} wave iterations, alive lanes {st["live_lane_iterations"] / (64 * st["wave_iterations"]):.3f}, kernel {R.last_kernel_ms(b) if hasattr(R, "last_kernel_ms") else 0} ms')
for n, c in zip(names, cyc): print(f'    {n:18s} {c / tot * 100:6.2f} %   {c / st["wave_iterations"]:10.0f} wave-cycles per iteration')
