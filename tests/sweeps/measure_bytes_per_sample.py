"""Algorithmic bytes per sample (path events x f32-compact record sizes) of every BASELINE workload, from the CPU
oracle's event counters on the workload's OWN pixel grid under the default seed.  The numbers go into
raytracinginrust_amd/workloads.py (BYTES_PER_SAMPLE) and BASELINE.md; bench.py uses those constants for `roofline.achieved` at every N.
Uses the oracle, so it lives under tests/.   usage: python tests/sweeps/measure_bytes_per_sample.py [C1 C2 ...]
spp of the measurement: 16 (C5: 4, same scene as C2 on a 13x larger grid); the statistical error of the mean is < 0.1 %."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from raytracinginrust_amd import scenes, workloads  # noqa: E402

SPP = {"C1": 64, "C2": 16, "C3": 16, "C4": 16, "C5": 4}

if __name__ == "__main__":
    be = orc.load()
    earth = scenes.load_earthmap()
    for key in (sys.argv[1:] or list(workloads.WORKLOADS)):
        w = workloads.WORKLOADS[key]
        b, cam, bg = workloads.build(w, be, earth)
        spp = SPP[key]
        t = time.perf_counter()
        _, cnt = orc.render(b, cam, bg, w.W, w.H, spp, w.max_depth, want_counters=True)
        dt = time.perf_counter() - t
        per_event = {k: cnt[k] * v / cnt["samples"] for k, v in orc.RECORD_BYTES.items() if cnt[k]}
        # the framebuffer term is amortised over the workload's real spp, not the measurement's
        bps = sum(per_event.values()) + orc.FRAMEBUFFER_BYTES_PER_PIXEL / w.spp
        print(f"{key}: {bps:.1f} B/sample  ({w.describe()}; measured on {w.W}x{w.H} at {spp} spp = {cnt['samples'] / 1e6:.1f} M samples, {dt:.1f} s; "
              f"bounces/sample {cnt['bounces'] / cnt['samples']:.3f}; " + ", ".join(f"{k} {v:.1f}" for k, v in per_event.items()) + ")", flush=True)
