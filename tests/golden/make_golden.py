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

if __name__ == "__main__":
    earth = scenes.load_earthmap()          # the reference's own 1024x512 texture, decoded by the library's JPEG ingest
    be = orc.load()
    for name, (W, H, spp, depth) in CASES.items():
        b, cam, bg = build_scene(name, be, earth)
        out, cnt = orc.render(b, cam, bg, W, H, spp, depth, seed=scenes.DEFAULT_SEED, want_counters=True)
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), rgb_sum=out, W=W, H=H, spp=spp, depth=depth,
                            seed=scenes.DEFAULT_SEED, bytes_per_sample=orc.algorithmic_bytes_per_sample(cnt, spp))
        print(name, out.shape, float(out.mean()) / spp, orc.algorithmic_bytes_per_sample(cnt, spp))
