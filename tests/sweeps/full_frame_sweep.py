#!/usr/bin/env python3
"""Whole frames of the BASELINE configs at their FULL sample count, every pixel against the CPU oracle (round 6; not in the suite: the
oracle side of C2 is ~40 s on 16 threads, C3 and C4 several minutes, C5 hours — `--minutes` bounds it to as many rows as that buys,
spread over the frame in bands).  The suite's test_full_spp_band_against_the_oracle compares 9 / 3 / 3 / 2 rows per config; this is the
same comparison over everything: the GPU frame through ONE launch (what bench.py times), the oracle's per-pixel sums of the same
(pixel, sample) streams, tolerance per pixel 1e-9 * (spp + |ref|), non-finite pixels equal.

    python3 tests/sweeps/full_frame_sweep.py C2 [C3 C4 C5] [--minutes M] [--threads T] [--out gpurun_out/full_frame.json]

Prints one line per config and writes a JSON record (copy it into profiles/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SAMPLE_RTOL = 1e-9      # tests/test_parity_gpu.py


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("keys", nargs="+")
    ap.add_argument("--minutes", type=float, default=0.0, help="bound on the oracle's wall time per config (0: the whole frame)")
    ap.add_argument("--threads", type=int, default=0, help="oracle threads (0: what this process may use)")
    ap.add_argument("--bands", type=int, default=8, help="with --minutes: the rows that fit are spread over this many bands")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_frame_sweep.json"))
    a = ap.parse_args()

    from raytracinginrust_amd import _lib, buildinfo, render as R, scenes, workloads
    from oracle import orc
    import bench
    pbe, obe = _lib.load(), orc.load_nocount()
    threads = a.threads or bench.usable_cores()
    earth = None
    record = {"kernel_source_id": buildinfo.kernel_source_id(), "oracle": orc.BUILD_INFO["compiler"], "threads": threads,
              "tolerance": "per pixel and channel |gpu - oracle| <= 1e-9 * (spp + |oracle|); non-finite pixels equal", "configs": {}}
    for key in a.keys:
        w = workloads.WORKLOADS[key]
        if w.scene == "final" and earth is None:
            earth = scenes.load_earthmap()
        b, cam, bg = workloads.build(w, pbe, earth)
        t = time.perf_counter()
        got = R.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth)
        gpu_s = time.perf_counter() - t
        k_ms = R.last_kernel_ms(b)
        ob, ocam, obg = workloads.build(w, obe, earth)
        # which rows: everything, or what --minutes buys at the rate of a 1-row probe in the middle of the frame, in bands
        rows = [(0, w.H)]
        if a.minutes > 0:
            t = time.perf_counter()
            probe = min(w.H, 2 * threads)                 # (the oracle's threads share ROWS: a one-row probe would time one thread)
            orc.render(ob, ocam, obg, w.W, w.H, w.spp, w.max_depth, nthreads=threads, mode=0, rows=(w.H // 2, w.H // 2 + probe))
            per_row = (time.perf_counter() - t) / probe
            n_rows = int(min(w.H, max(a.bands, a.minutes * 60.0 / per_row)))
            if n_rows < w.H:
                per = max(1, n_rows // a.bands)
                starts = [int(round(i * (w.H - per) / max(1, a.bands - 1))) for i in range(a.bands)]
                rows = sorted(set((s, s + per) for s in starts))
        ref = np.zeros((w.H, w.W, 3))
        mask = np.zeros(w.H, dtype=bool)
        t = time.perf_counter()
        for r0, r1 in rows:
            part = orc.render(ob, ocam, obg, w.W, w.H, w.spp, w.max_depth, nthreads=threads, mode=0, rows=(r0, r1))
            ref[r0:r1] = part[r0:r1]
            mask[r0:r1] = True
        cpu_s = time.perf_counter() - t
        g, r = got[mask], ref[mask]
        fin = np.isfinite(r)
        same_nonfinite = bool(np.array_equal(np.isfinite(g), fin))
        d = np.abs(np.where(fin, g, 0.0) - np.where(fin, r, 0.0))
        bad = (d > SAMPLE_RTOL * (w.spp + np.abs(np.where(fin, r, 0.0)))).any(axis=-1)
        ok = ~bad
        equal_words = float((g.view(np.uint64) == r.view(np.uint64)).all(axis=-1).mean())
        rec = {"workload": w.describe(), "rows_compared": int(mask.sum()), "rows_of_frame": w.H, "row_ranges": [list(x) for x in rows],
               "pixels_compared": int(mask.sum()) * w.W, "samples_compared": int(mask.sum()) * w.W * w.spp,
               "pixels_off": int(bad.sum()), "nonfinite_pixels_equal": same_nonfinite, "nonfinite_pixels": int((~fin).any(axis=-1).sum()),
               "max_abs_diff_of_the_rest": float(d[ok].max()) if ok.any() else None, "max_pixel_sum": float(np.abs(r[fin]).max()),
               "worst_diff": float(d.max()), "pixels_bit_identical": equal_words,
               "gpu_kernel_ms": k_ms, "gpu_call_s": gpu_s, "oracle_s": cpu_s,
               "oracle_Msamples_per_s": int(mask.sum()) * w.W * w.spp / cpu_s / 1e6, "gpu_Msamples_per_s": w.samples / (k_ms * 1e-3) / 1e6}
        record["configs"][key] = rec
        print(f"{key} {w.describe()}: rows {rec['rows_compared']} / {w.H} ({rec['samples_compared'] / 1e9:.2f} G samples), {rec['pixels_off']} pixels off, "
              f"non-finite equal: {same_nonfinite}, max |gpu - oracle| of the rest {rec['max_abs_diff_of_the_rest']:.3e} (sums up to {rec['max_pixel_sum']:.1f}), "
              f"{equal_words:.1%} of the pixel sums bit-identical; GPU {k_ms:.1f} ms, oracle {cpu_s:.0f} s on {threads} threads", flush=True)
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(record, open(a.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
