#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the path-tracing hot path on BASELINE config 2
(Cornell box 800x800, 1024 spp, depth 50) in f64 (the reference's arithmetic type).

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole frame: every rank renders its interleaved tiles (one persistent HIP kernel launch per
rank), then ONE gather (RCCL over xGMI) moves them to rank 0.  The frame is fixed as N grows (strong scaling).
For N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints one JSON line.  `roofline.achieved` = algorithmic bytes per launch (SURVEY §8(d) model x samples in
the launch) / mean kernel duration from HIP events recorded on the launch stream inside the timed region.
`cpu_baseline` (N = 1 only) times the CPU oracle — a restatement of the reference, not the Rust binary, which
cannot be built here — on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def usable_cores():
    """Threads this process can really run at once: hardware threads, CPU affinity and the cgroup CPU quota (a GPU box's
    container sees every hardware thread of the host but is throttled to its share)."""
    from oracle import orc
    n = min(orc.hardware_threads(), len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                 # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())               # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(w, sample_spp, earth=None):
    """CPU restatement of the reference (oracle/), all host threads, on the workload's own pixel grid.  `sample_spp` > 0 fixes
    the samples per pixel of the bounded sample; -1 sizes it for about 8 s of wall time from a 1-spp calibration pass.
    Also returns the oracle's algorithmic bytes/sample for this workload."""
    from oracle import orc
    from raytracinginrust_amd import workloads
    be = orc.load()
    b, cam, bg = workloads.build(w, be, earth)
    threads = usable_cores()
    if sample_spp < 0:
        t = time.perf_counter()
        orc.render(b, cam, bg, w.W, w.H, 1, w.max_depth, nthreads=threads, mode=0)
        per_spp = max(time.perf_counter() - t, 1e-3)
        sample_spp = int(min(256, max(4, round(8.0 / per_spp))))
    t = time.perf_counter()
    _, cnt = orc.render(b, cam, bg, w.W, w.H, sample_spp, w.max_depth, want_counters=True, nthreads=threads, mode=0)
    dt = time.perf_counter() - t
    n = w.W * w.H * sample_spp
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{w.scene} {w.W}x{w.H} at {sample_spp} spp ({n / 1e6:.1f} Msamples, {dt:.1f} s wall, "
                      f"oracle f64, threads over rows)"}, orc.algorithmic_bytes_per_sample(cnt, sample_spp)


def pmc_traffic_bytes(workload_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (separate FETCH_SIZE and
    WRITE_SIZE runs; counters are in KB; raw values — the accesses are 8-byte f64 atomics, for which the guide has no
    correction).  None if no profile of this workload is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_bench_{workload_key}_pmc_summary.csv")))
    if not files:
        return None, None, None
    vals = {r["counter"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(files[-1]))}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None, None
    # secondary figure: fraction of SIMD time the vector ALU is busy.  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES
    # both count quad-cycles summed over waves; SQ_WAVES / 1024 SIMDs = resident waves per SIMD of the persistent grid.
    valu = None
    if all(k in vals for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES")) and vals["SQ_WAVE_CYCLES"] > 0:
        valu = vals["SQ_ACTIVE_INST_VALU"] * (vals["SQ_WAVES"] / 1024.0) / vals["SQ_WAVE_CYCLES"]
    return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, os.path.basename(files[-1]), valu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--tile-px", type=int, default=67, help="pixels per tile; prime by default (see dist.DEFAULT_TILE_PX)")
    ap.add_argument("--f32", action="store_true", help="throughput variant (not the headline: reduced precision)")
    ap.add_argument("--near-first", action="store_true", help="opt-in RT_NEAR_FIRST_BVH traversal (not the reference's order)")
    ap.add_argument("--pipeline", type=int, default=1, choices=(1, 2),
                    help="frames in flight; 2 lets a frame's drain overlap the next frame's start on a second stream (measured: -3 %% per "
                         "1/8-frame share when the two streams land on different hardware queues, nothing otherwise; off by default)")
    ap.add_argument("--sah", action="store_true", help="opt-in RT_BVH_SAH builder (not the reference's tree shape)")
    ap.add_argument("--cpu-spp", type=int, default=-1, help="spp of the bounded CPU-baseline sample (0 = skip, -1 = sized for ~8 s of wall time)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracinginrust_amd import _lib, dist as D, render as R, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    # RT_BENCH_BACKEND=gloo lets the N > 1 flow be exercised with several ranks sharing one GPU (development boxes
    # have a single GPU, and RCCL refuses two ranks on one device); the driver's runs use the default, nccl = RCCL.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    be = _lib.load()            # after `import torch`: one HIP runtime in the process
    w = workloads.WORKLOADS[args.workload]
    earth = None
    if w.scene == "final":
        from raytracinginrust_amd import scenes
        earth = scenes.load_earthmap()
    b, cam, bg = workloads.build(w, be, earth)
    flags = (R.RT_F32 if args.f32 else R.RT_F64) | (R.RT_NEAR_FIRST_BVH if args.near_first else 0)
    if args.sah:
        R.set_bvh_builder(b, R.RT_BVH_SAH)
    tr = D.TileRenderer(b, cam, bg, w.W, w.H, w.spp, w.max_depth, flags=flags, tile_px=args.tile_px, rank=rank, world=world,
                        pipeline=args.pipeline)

    def sync():
        torch.cuda.synchronize()            # every stream of this rank (frames run on two alternating side streams)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time initialisation (scene flatten + upload, code-object load, event creation) is not part of a step: do it now
    # (no kernel launch), so that timed steps measure the hot path even with --warmup 0
    R.prepare(b, flags)
    if world > 1:
        # communicators and point-to-point channels are created lazily on first use: do that outside the timed region
        # even when --warmup 0 is requested
        tiny = torch.zeros(8, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.gather(tiny, [torch.empty_like(tiny) for _ in range(world)] if rank == 0 else None, dst=0)
        dist.all_reduce(tiny)
    for _ in range(args.warmup):
        tr.render_frame(dst=0)
    sync()
    R.kernel_time_total(b, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = tr.render_frame(dst=0)             # asynchronous: no host stop between frames
    sync()
    elapsed = time.perf_counter() - t0
    # HIP events around every kernel on its own launch stream, summed by the library (no host stop after each frame)
    k_total_ms, k_launches = R.kernel_time_total(b)
    assert k_launches == args.steps
    stats = R.last_stats(b)
    n_flush = R.last_flush_count(b)
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        assert frame is not None and tuple(frame.shape) == (w.H, w.W, 3)
        mean_radiance = float(torch.nan_to_num(frame).mean().item()) / w.spp
        value = w.samples * args.steps / elapsed / 1e6
        cpu, bps = None, workloads.BYTES_PER_SAMPLE.get(w.key)
        if world == 1 and args.cpu_spp != 0 and not args.f32:
            cpu, bps = cpu_baseline(w, args.cpu_spp, earth)
        k_ms = k_total_ms / k_launches
        n_px = w.W * w.H                     # real (unpadded) pixels rank 0's launch owns
        local_px = sum(max(0, min(n_px, (t + 1) * args.tile_px) - t * args.tile_px)
                       for t in D.local_tile_ids(w.W, w.H, args.tile_px, rank, world) if t * args.tile_px < n_px)
        local_samples = local_px * w.spp
        roof = None
        if bps is not None:
            achieved = bps * local_samples / (k_ms * 1e-3) / 1e9
            traffic, traffic_src, valu_busy = pmc_traffic_bytes(w.key) if (world == 1 and not args.f32) else (None, None, None)
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": traffic, "traffic_unit": "bytes per launch (PMC FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": bps * local_samples, "kernel": f"rt::pathtrace_kernel<{'float' if args.f32 else 'double'}, FEATS> (FEATS = 0 for the Cornell box; the leanest instantiation covering the scene)",
                    "kernel_ms": k_ms, "bytes_per_sample": bps,
                    "valu_busy_frac": valu_busy,        # from the same committed PMC passes: what actually limits the kernel
                    "framebuffer_atomics_per_launch": 3 * n_flush,       # what the kernel itself counted: f64 atomic adds, 8 B each
                    "framebuffer_atomic_bytes_per_launch": 24 * n_flush,
                    "note": "algorithmic bytes (event x record-size model, SURVEY 8(d)); the scene is L2/LDS-resident, "
                            "physical HBM traffic is ~ the framebuffer (see DESIGN.md / profiles/)"}
        out = {
            "metric": "Msamples/s (pixels x spp / s)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64", "data": "synthetic",
            "config": {"workload": f"{w.key}: {w.describe()}", "tile_px": args.tile_px, "seed": "0x5EED",
                       "parallelism": f"tiles interleaved over {world} GPU(s) + 1 gather",
                       "frames_in_flight": tr.pipeline},
            "roofline": roof, "cpu_baseline": cpu,
            "lane_utilisation": stats["live_lane_iterations"] / max(1, 64 * stats["wave_iterations"]),
            "nonfinite_samples_rank0": stats["nonfinite_samples"], "mean_radiance": mean_radiance,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
