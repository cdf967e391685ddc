"""Per-scene parity report (GPU vs CPU oracle, per sample) and kernel timings for the four BASELINE scenes; writes PNG
previews under gpurun_out/.  Developer tool: run on the GPU box with `python tests/sweeps/parity_report.py`."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from raytracinginrust_amd import _lib, scenes, render as R
from oracle import orc
from PIL import Image
obe = orc.load(); be = _lib.load()
eb, ew, eh = scenes.load_earthmap()
def build(name, backend):
    if name == 'cornell': return scenes.cornell_box(backend)
    if name == 'random': return scenes.random_scene(backend, aspect_ratio=16/9)
    if name == 'final': return scenes.final_scene(backend, eb, ew, eh)
    if name == 'teapot': return scenes.cornell_test(backend, scenes.asset_path('teapot.obj'))
def compare(name, W, H, spp, depth, flags=0):
    ob, ocam, obg = build(name, obe); pb, pcam, pbg = build(name, be)
    print(name, R.flatten(pb))
    t=time.time(); ref, rs, cnt = orc.render(ob, ocam, obg, W, H, spp, depth, want_samples=True, want_counters=True); to=time.time()-t
    t=time.time(); got, gs = R.render(pb, pcam, pbg, W, H, spp, depth, want_samples=True, flags=flags); tg=time.time()-t
    st = R.last_stats(pb)
    d = np.abs(got-ref); ds = np.abs(gs-rs)
    fin = np.isfinite(rs).all(axis=-1) & np.isfinite(gs).all(axis=-1)
    nanmatch = (np.isfinite(rs) == np.isfinite(gs)).all()
    tol = 1e-9*(1+np.abs(rs))
    nbad = ((ds > tol) & np.isfinite(ds)).any(axis=-1).sum()
    print(f'  {name}: oracle {to:.2f}s gpu {tg:.3f}s kernel {R.last_kernel_ms(pb):.2f}ms | pixel max abs {np.nanmax(d):.3e} | samples: max abs {np.nanmax(ds):.3e} bad {nbad}/{W*H*spp} bit-identical {(gs==rs).all(axis=-1).mean():.4f} nonfinite orc {cnt["nonfinite"]} gpu {st["nonfinite_samples"]} nanmatch {nanmatch} B/sample {orc.algorithmic_bytes_per_sample(cnt, spp):.0f}')
    if nbad:
        idx = np.argwhere(((ds > tol)).any(axis=-1))[:5]
        for i in idx: print('   bad at', i, rs[tuple(i)], gs[tuple(i)])
    return got
for name, W, H, spp, depth in [('cornell', 48, 48, 16, 50), ('random', 64, 36, 16, 8), ('final', 48, 48, 16, 50), ('teapot', 64, 36, 16, 50)]:
    try: compare(name, W, H, spp, depth)
    except Exception as e:
        import traceback; traceback.print_exc()
# timing of the full-feature scenes
for name, W, H, spp, depth in [('cornell', 800, 800, 256, 50), ('random', 400, 225, 64, 8), ('final', 400, 400, 64, 50), ('teapot', 480, 270, 64, 50)]:
    pb, pcam, pbg = build(name, be)
    for flags in (0, 1):
        got = R.render(pb, pcam, pbg, W, H, spp, depth, flags=flags)
        ms = R.last_kernel_ms(pb); st = R.last_stats(pb)
        print(f'{name} {W}x{H}x{spp} flags={flags}: kernel {ms:.2f} ms  {W*H*spp/ms/1e3:.1f} Msamples/s  lane util {st["live_lane_iterations"]/(64*st["wave_iterations"]):.3f} nonfinite {st["nonfinite_samples"]} mean {np.nanmean(got)/spp:.4f}')
    img = np.clip(np.sqrt(np.nan_to_num(got/spp)), 0, 0.999)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    Image.fromarray((img*256).astype(np.uint8)).save(os.path.join(ROOT, 'gpurun_out', f'{name}.png'))
