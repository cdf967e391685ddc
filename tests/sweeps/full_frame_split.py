#!/usr/bin/env python3
"""full_frame_sweep.py for a frame whose oracle side outlasts a GPU-box call (C5: 66 minutes on 16 threads): the same comparison —
the GPU frame through ONE launch (what bench.py times) against the oracle's per-pixel sums of the same (pixel, sample) streams,
tolerance per pixel 1e-9 * (spp + |ref|), non-finite pixels equal — with the two sides computed where each can be:

    oracle  (any host, no GPU; resumable: one .npy per chunk of rows, finished chunks are skipped)
        python3 tests/sweeps/full_frame_split.py oracle C5 --skip profiles/r06_full_frame_C5.json --dir gpurun_out/c5_oracle --threads 6
    gpu     (GPU box: renders the whole frame once, keeps the rows asked for; <= 64 MiB per call travel back, 700 rows of C5)
        python3 tests/sweeps/full_frame_split.py gpu C5 --rows 55:140,195:281 --out gpurun_out/c5_gpu_a.npz
    compare (any host: every row that both sides hold)
        python3 tests/sweeps/full_frame_split.py compare C5 --dir gpurun_out/c5_oracle --gpu gpurun_out/c5_gpu_a.npz gpurun_out/c5_gpu_b.npz \
            --out profiles/r06_full_frame_C5_rest.json

`--skip` takes the row ranges an earlier full_frame_sweep.py record already covers and leaves them out.  Not in the suite."""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SAMPLE_RTOL = 1e-9      # tests/test_parity_gpu.py


def parse_rows(s):
    return [tuple(int(x) for x in part.split(":")) for part in s.split(",") if part]


def complement(H, done):
    """Row ranges of [0, H) that `done` (sorted, disjoint [r0, r1) pairs) leaves out."""
    out, at = [], 0
    for r0, r1 in sorted(done):
        if r0 > at:
            out.append((at, r0))
        at = max(at, r1)
    if at < H:
        out.append((at, H))
    return out


def wanted_rows(a, w):
    if a.rows:
        return parse_rows(a.rows)
    if a.skip:
        rec = json.load(open(a.skip))["configs"][w.key]
        return complement(w.H, [tuple(x) for x in rec["row_ranges"]])
    return [(0, w.H)]


def cmd_oracle(a, w):
    from raytracinginrust_amd import scenes, workloads
    from oracle import orc
    be = orc.load_nocount()
    earth = scenes.load_earthmap() if w.scene == "final" else None
    b, cam, bg = workloads.build(w, be, earth)
    os.makedirs(a.dir, exist_ok=True)
    chunks = [(r, min(r + a.chunk, r1)) for r0, r1 in wanted_rows(a, w) for r in range(r0, r1, a.chunk)]
    t_all = time.perf_counter()
    for i, (r0, r1) in enumerate(chunks):
        path = os.path.join(a.dir, f"{w.key}_rows_{r0:05d}_{r1:05d}.npy")
        if os.path.exists(path):
            continue
        t = time.perf_counter()
        part = orc.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth, nthreads=a.threads, mode=0, rows=(r0, r1))
        np.save(path + ".tmp.npy", part[r0:r1])
        os.replace(path + ".tmp.npy", path)
        print(f"{w.key} rows {r0}..{r1} ({i + 1} / {len(chunks)}): {time.perf_counter() - t:.0f} s on {a.threads} threads, "
              f"{(time.perf_counter() - t_all) / 60:.1f} min so far", flush=True)
    return 0


def cmd_gpu(a, w):
    from raytracinginrust_amd import _lib, buildinfo, render as R, scenes, workloads
    be = _lib.load()
    earth = scenes.load_earthmap() if w.scene == "final" else None
    b, cam, bg = workloads.build(w, be, earth)
    got = R.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth)          # ONE launch over the whole frame, as bench.py's step
    rows = wanted_rows(a, w)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez(a.out, ranges=np.asarray(rows, dtype=np.int64), kernel_ms=R.last_kernel_ms(b), kernel_source_id=buildinfo.kernel_source_id(),
             **{f"rows_{r0}_{r1}": got[r0:r1] for r0, r1 in rows})
    print(f"{w.key} {w.describe()}: one launch, {R.last_kernel_ms(b):.1f} ms; kept {sum(r1 - r0 for r0, r1 in rows)} rows in {a.out}", flush=True)
    return 0


def cmd_compare(a, w):
    from oracle import orc
    gpu = np.full((w.H, w.W, 3), np.nan)
    have_gpu = np.zeros(w.H, dtype=bool)
    k_ms, ident = [], set()
    for path in a.gpu:
        z = np.load(path)
        k_ms.append(float(z["kernel_ms"]))
        ident.add(str(z["kernel_source_id"]))
        for r0, r1 in z["ranges"]:
            gpu[r0:r1] = z[f"rows_{r0}_{r1}"]
            have_gpu[r0:r1] = True
    ref = np.zeros((w.H, w.W, 3))
    have_ref = np.zeros(w.H, dtype=bool)
    for path in sorted(glob.glob(os.path.join(a.dir, f"{w.key}_rows_*_*.npy"))):
        if path.endswith(".tmp.npy"):
            continue
        r0, r1 = (int(x) for x in os.path.basename(path)[:-4].split("_")[-2:])
        ref[r0:r1] = np.load(path)
        have_ref[r0:r1] = True
    mask = have_gpu & have_ref
    g, r = gpu[mask], ref[mask]
    fin = np.isfinite(r)
    same_nonfinite = bool(np.array_equal(np.isfinite(g), fin))
    d = np.abs(np.where(fin, g, 0.0) - np.where(fin, r, 0.0))
    bad = (d > SAMPLE_RTOL * (w.spp + np.abs(np.where(fin, r, 0.0)))).any(axis=-1)
    ok = ~bad
    idx = np.flatnonzero(mask)
    ranges, start = [], None
    for i, row in enumerate(idx):
        if start is None:
            start = row
        if i + 1 == len(idx) or idx[i + 1] != row + 1:
            ranges.append([int(start), int(row) + 1])
            start = None
    rec = {"workload": w.describe(), "rows_compared": int(mask.sum()), "rows_of_frame": w.H, "row_ranges": ranges,
           "pixels_compared": int(mask.sum()) * w.W, "samples_compared": int(mask.sum()) * w.W * w.spp,
           "pixels_off": int(bad.sum()), "nonfinite_pixels_equal": same_nonfinite, "nonfinite_pixels": int((~fin).any(axis=-1).sum()),
           "max_abs_diff_of_the_rest": float(d[ok].max()) if ok.any() else None, "max_pixel_sum": float(np.abs(r[fin]).max()),
           "worst_diff": float(d.max()), "pixels_bit_identical": float((g.view(np.uint64) == r.view(np.uint64)).all(axis=-1).mean()),
           "gpu_kernel_ms_per_whole_frame_launch": k_ms, "rows_only_on_one_side": int((have_gpu ^ have_ref).sum())}
    record = {"kernel_source_id": sorted(ident), "oracle": orc.BUILD_INFO["compiler"],
              "how": "tests/sweeps/full_frame_split.py: the GPU side is one whole-frame launch per call on the GPU box (the rows kept travel back), "
                     "the oracle side ran in the build container on the same oracle build; compared here",
              "tolerance": "per pixel and channel |gpu - oracle| <= 1e-9 * (spp + |oracle|); non-finite pixels equal", "configs": {w.key: rec}}
    print(f"{w.key} {w.describe()}: rows {rec['rows_compared']} / {w.H} ({rec['samples_compared'] / 1e9:.2f} G samples), {rec['pixels_off']} pixels off, "
          f"non-finite equal: {same_nonfinite}, max |gpu - oracle| of the rest {rec['max_abs_diff_of_the_rest']:.3e} (sums up to {rec['max_pixel_sum']:.1f}), "
          f"{rec['pixels_bit_identical']:.1%} of the pixel sums bit-identical", flush=True)
    if a.out:
        json.dump(record, open(a.out, "w"), indent=1)
    return 0 if rec["pixels_off"] == 0 and same_nonfinite else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=("oracle", "gpu", "compare"))
    ap.add_argument("key")
    ap.add_argument("--rows", default="", help="r0:r1[,r0:r1...] (output rows, end exclusive)")
    ap.add_argument("--skip", default="", help="a full_frame_sweep.py record: the rows it covers are left out")
    ap.add_argument("--dir", default=os.path.join(ROOT, "gpurun_out", "full_frame_oracle"), help="oracle chunks (one .npy per chunk of rows)")
    ap.add_argument("--chunk", type=int, default=24, help="rows per oracle chunk")
    ap.add_argument("--threads", type=int, default=0, help="oracle threads (0: every hardware thread)")
    ap.add_argument("--gpu", nargs="*", default=[], help="compare: the .npz files of the gpu calls")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from raytracinginrust_amd import workloads
    w = workloads.WORKLOADS[a.key]
    return {"oracle": cmd_oracle, "gpu": cmd_gpu, "compare": cmd_compare}[a.cmd](a, w)


if __name__ == "__main__":
    sys.exit(main())
