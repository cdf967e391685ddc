"""Generates tests/golden/oracle_*.npz: per-pixel sums of the CPU oracle for the four BASELINE scenes at
small sizes under the default seed.  The reference itself has no golden vectors and cannot run here
(Rust, no toolchain), so these pin the ORACLE's behaviour over time (a change to oracle.cpp that moves
them must be deliberate) and give the GPU tests a fixture that does not need the oracle library."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import build_scene  # noqa: E402
from oracle import orc  # noqa: E402
from raytracinginrust_amd import scenes  # noqa: E402

CASES = {  # name: (W, H, spp, depth)
    "cornell": (32, 32, 16, 50),
    "random": (32, 18, 8, 8),
    "final": (24, 24, 8, 50),
    "teapot": (32, 18, 8, 50),
}

# BASELINE configs 2-5 on their own pixel grids at the first k samples of every pixel (the full config's RNG streams 0..k-1): sums over
# 16x16-pixel blocks of the oracle's per-pixel sums (the frames themselves are 15-199 MB), the non-finite count per block.
GRID_SPP = {"C2": 16, "C3": 4, "C4": 2, "C5": 1}


def block_sums(img, blk=16):
    H, W, _ = img.shape
    fin = np.isfinite(img).all(axis=-1)
    v = np.where(fin[..., None], img, 0.0)
    ys, xs = np.arange(0, H, blk), np.arange(0, W, blk)
    return (np.add.reduceat(np.add.reduceat(v, ys, axis=0), xs, axis=1),
            np.add.reduceat(np.add.reduceat((~fin).astype(np.int64), ys, axis=0), xs, axis=1))


def make_grid_goldens(be, earth):
    from raytracinginrust_amd import workloads
    for key, spp in GRID_SPP.items():
        w = workloads.WORKLOADS[key]
        b, cam, bg = workloads.build(w, be, earth)
        out = orc.render(b, cam, bg, w.W, w.H, spp, w.max_depth, seed=scenes.DEFAULT_SEED)
        sums, bad = block_sums(out)
        np.savez_compressed(os.path.join(HERE, f"oracle_grid_{key}.npz"), block_sums=sums, nonfinite=bad, W=w.W, H=w.H, spp=spp, depth=w.max_depth,
                            seed=scenes.DEFAULT_SEED)
        print(key, sums.shape, float(sums.sum()) / (w.W * w.H * spp), int(bad.sum()))


# BASELINE configs 2-5 at their FULL sample counts on two bands of rows of their own frames: a band in the middle of the image and the
# top row (output row 0).  Per-pixel sums of the oracle — what links samples 16..1023 / 4..4095 / 2..2047 / 1..8191 of every pixel, the
# 256-sample tail chunks and the pixel x spp > 2^32 bookkeeping of the product's full-frame launch to the oracle per pixel.
BAND_ROWS = {"C2": 8, "C3": 2, "C4": 2, "C5": 1}


def band_rows(key, H):
    n = BAND_ROWS[key]
    mid = H // 2 - n // 2
    return [(mid, mid + n), (0, 1)]


def make_band_goldens(be, earth, keys=None):
    import time
    from raytracinginrust_amd import workloads
    for key in keys or BAND_ROWS:
        w = workloads.WORKLOADS[key]
        b, cam, bg = workloads.build(w, be, earth)
        rows, sums = [], []
        t0 = time.perf_counter()
        for r0, r1 in band_rows(key, w.H):
            out = orc.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth, seed=scenes.DEFAULT_SEED, rows=(r0, r1))
            rows.append((r0, r1)); sums.append(out[r0:r1].copy())
        np.savez_compressed(os.path.join(HERE, f"oracle_band_{key}.npz"), rows=np.array(rows), band0=sums[0], band1=sums[1], W=w.W, H=w.H, spp=w.spp,
                            depth=w.max_depth, seed=scenes.DEFAULT_SEED)
        print(key, rows, [float(np.nanmean(x)) / w.spp for x in sums], [int((~np.isfinite(x)).sum()) for x in sums], f"{time.perf_counter() - t0:.1f} s")


if __name__ == "__main__":
    earth = scenes.load_earthmap()          # the reference's own 1024x512 texture, decoded by the library's JPEG ingest
    be = orc.load()
    for name, (W, H, spp, depth) in CASES.items():
        b, cam, bg = build_scene(name, be, earth)
        out, cnt = orc.render(b, cam, bg, W, H, spp, depth, seed=scenes.DEFAULT_SEED, want_counters=True)
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), rgb_sum=out, W=W, H=H, spp=spp, depth=depth,
                            seed=scenes.DEFAULT_SEED, bytes_per_sample=orc.algorithmic_bytes_per_sample(cnt, spp))
        print(name, out.shape, float(out.mean()) / spp, orc.algorithmic_bytes_per_sample(cnt, spp))
    make_grid_goldens(be, earth)
    make_band_goldens(orc.load_nocount() if hasattr(orc, 'load_nocount') else be, earth)
