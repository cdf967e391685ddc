"""Kernel time of the BASELINE workloads at reduced spp (same frame sizes): python tools/workloads_time.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from PIL import Image
from raytracinginrust_amd import _lib, render as R, scenes, workloads
be = _lib.load()
earth = scenes.load_earthmap()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for key in os.environ.get('RT_WORKLOADS', 'C1,C2,C3,C4').split(','):
    w = workloads.WORKLOADS[key]
    b, cam, bg = workloads.build(w, be, earth)
    s = min(spp, w.spp)
    if os.environ.get('RT_BVH'): R.set_bvh_builder(b, int(os.environ['RT_BVH']))
    if os.environ.get('RT_TRAV'): R.set_traversal_schedule(b, *[int(x) for x in os.environ['RT_TRAV'].split(',')])
    flags = int(os.environ.get('RT_FLAGS', '0'))
    for _ in range(2):
        out = R.render(b, cam, bg, w.W, w.H, s, w.max_depth, flags=flags)
    ms = R.last_kernel_ms(b); st = R.last_stats(b)
    tv = R.last_traversal_stats(b)
    trav = f"  adv {tv['advance_lanes']/max(1,64*tv['advance_passes']):.2f} x {tv['advance_passes']/1e6:.2f}M  trav {tv['traversal_lanes']/max(1,64*tv['traversal_steps']):.2f} x {tv['traversal_steps']/1e6:.1f}M" if tv['traversal_steps'] else ''
    print('   ', R.last_launch_info(b))
    print(f'{key} {w.scene} {w.W}x{w.H} at {s} spp: {ms:9.2f} ms  {w.W*w.H*s/ms/1e3:8.1f} Msamples/s  lane util {st["live_lane_iterations"]/max(1,64*st["wave_iterations"]):.3f}  -> full config ({w.spp} spp) ~ {ms*w.spp/s/1e3:.2f} s{trav}')
