#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the path-tracing hot path on BASELINE config 2
(Cornell box 800x800, 1024 spp, depth 50) in f64 (the reference's arithmetic type).

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole frame: every GPU renders its interleaved tiles (one persistent HIP kernel launch per GPU), then ONE gather
(RCCL over xGMI) moves them to the first GPU, which un-permutes them into the frame.  The frame is fixed as N grows (strong scaling).
Three ways to run it:
  * N = 1: `python bench.py` — rt_render_device on the one GPU;
  * N > 1 under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, WORLD_SIZE set): one
    process per GPU, rt_render_device per rank + one torch.distributed.gather (`--mode procs`);
  * N > 1 WITHOUT a launcher (`python bench.py --gpus N`): this one process drives the N GPUs through rt_render_multi_device — what
    the reference's single-process `main` (src/main.rs:767-835) would call: ncclCommInitAll + one ncclGather + un-permute on the
    device (`--mode inproc`).  On a box with fewer than N GPUs this mode needs the test hook RT_MULTI_VIRTUAL_RANKS=N (N ranks on
    the one device; the line says so) — it never re-launches or execs anything.
Rank 0 prints one JSON line.

`roofline`: the bound that is real for this path, f64 VALU issue.  `achieved` = the reference's f64 operations per sample (counted
by kind by the op-counting build of the CPU oracle on the workload's own grid, committed as workloads.F64_OPS_PER_SAMPLE, weighted
by what each kind costs in issue slots: workloads.VALU_OP_WEIGHTS) x the samples in one launch / that kernel's mean duration from
HIP events recorded on the launch stream inside the timed region; `peak` = 39.3 T lane-operations/s (256 CUs x 4 SIMDs x 16 f64
lanes x 2.4 GHz; the path has no FMAs).  BASELINE's own metric — algorithmic bytes (SURVEY 8(d) event x record-size model) over
the 8 TB/s HBM peak — is kept beside it as `roofline.model_hbm`; the scene is L2/LDS-resident, so that is a model figure, not HBM
utilisation, and it is flagged when it exceeds what HBM could deliver.  Physical traffic (`traffic`) and the PMC view of the VALU
(`valu_pmc`) come from the committed rocprofv3 --pmc passes of this same command and are used only when that profile was taken on
THIS build of the kernels and launch code (matching `kernel_source_id`); otherwise they are null.

`loop` (headline and every workload): what ran — loop shape, the instantiation's FEATS and name as a profiler shows it, how the shape was
chosen, the calibration's two kernel times (rt_last_loop_info).  Before anything is timed every workload's view is calibrated
(rt_scene_calibrate: mesh scenes measure their loop shape — rank 0's verdict is handed to every rank —, one-BVH worlds tune their filter
tree; `--loop persistent|lockstep` sets the shape instead, which is what the profiled runs of tools/profile_all.sh do).  PMC-derived fields
are replayed only from a committed profile of this build AND this instantiation; `roofline.frac` is reference-equivalent throughput,
`roofline.executed_valu_frac` what the SIMDs really issue.

`workloads`: the other BASELINE configs (C1, C3, C4, C5) timed the same way, one reduced-spp warm-up frame + one full frame each
(twenty for C1, whose frame is a few milliseconds), so that every config's Msamples/s and roofline fraction is on the driver's clock.

`cpu_baseline` (N = 1 only): the CPU oracle — a restatement of the reference, not the Rust binary, which cannot be built here —
on a bounded sample of the same workload, in both threading shapes BASELINE.md names, counters compiled out.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024               # 256 CUs x 4 SIMDs


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
    """CPU restatement of the reference (oracle/, built with its event counters compiled out), all usable host threads, on the
    workload's own pixel grid, in the two threading shapes BASELINE.md §2 names:
      rows              threads over image rows (dynamic)                         -> `value`
      reference_shaped  pixels sequential, the samples of ONE pixel split over the threads, as the reference's
                        `(0..SPP).into_par_iter()` does (src/main.rs:811), persistent worker pool
    `sample_spp` > 0 fixes the samples per pixel of the `rows` sample; -1 sizes it for about 8 s of wall time from a 1-spp
    calibration pass.  The reference-shaped sample is a band of rows in the middle of the frame at the workload's full spp, sized
    for about 4 s."""
    from oracle import orc
    from raytracinginrust_amd import workloads
    be = orc.load_nocount()
    b, cam, bg = workloads.build(w, be, earth)
    threads = usable_cores()
    t = time.perf_counter()
    orc.render(b, cam, bg, w.W, w.H, 1, w.max_depth, nthreads=threads, mode=0)
    per_spp = max(time.perf_counter() - t, 1e-3)
    if sample_spp < 0:
        sample_spp = int(min(256, max(4, round(8.0 / per_spp))))
    t = time.perf_counter()
    orc.render(b, cam, bg, w.W, w.H, sample_spp, w.max_depth, nthreads=threads, mode=0)
    dt = time.perf_counter() - t
    n = w.W * w.H * sample_spp
    # reference-shaped: rows [r0, r1) around the middle of the frame at full spp, about 4 s of work at the rate just measured
    rate = n / dt
    band = int(max(1, min(w.H, round(4.0 * rate / (w.W * w.spp)))))
    r0 = max(0, w.H // 2 - band // 2)
    t = time.perf_counter()
    orc.render(b, cam, bg, w.W, w.H, w.spp, w.max_depth, nthreads=threads, mode=1, rows=(r0, r0 + band))
    dt1 = time.perf_counter() - t
    n1 = band * w.W * w.spp
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{w.scene} {w.W}x{w.H} at {sample_spp} spp ({n / 1e6:.1f} Msamples, {dt:.1f} s wall, oracle f64, threads over rows)",
            "reference_shaped": {"value": n1 / dt1 / 1e6, "unit": "Msamples/s", "cores": threads,
                                 "sample": f"{w.scene} {w.W}x{w.H}, rows {r0}..{r0 + band} at the full {w.spp} spp ({n1 / 1e6:.1f} Msamples, "
                                           f"{dt1:.1f} s wall), pixels sequential, each pixel's samples split over the threads (src/main.rs:811)"},
            "compiler": orc.BUILD_INFO["compiler"], "flags": orc.BUILD_INFO["flags"],
            "note": "CPU restatement of the reference (oracle/), never the Rust binary; event counters compiled out"}


def pmc_profile(workload_key, kernel=None):
    """Per-dispatch counters of the TIMED frames from the committed rocprofv3 --pmc passes of this same command (separate FETCH_SIZE /
    WRITE_SIZE / SQ passes, tools/profile_pmc.sh + tools/pmc_summary.py), newest snapshot of this workload — used ONLY when it was
    taken on this build of the kernels (the summary's kernel_source_id equals the hash of the kernel sources in this tree) AND of the
    instantiation this run launched (`kernel`, from rt_last_loop_info: a mesh scene may run either loop shape).  Returns
    (values, file name) or (None, reason)."""
    import csv
    import glob
    from raytracinginrust_amd import buildinfo
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_bench_{workload_key}_pmc_summary.csv")))
    if not files:
        return None, "no PMC profile of this workload committed"
    rows = list(csv.DictReader(open(files[-1])))
    ident = [r["mean_per_dispatch"] for r in rows if r["counter"] == "kernel_source_id"]
    if not ident or ident[0] != buildinfo.kernel_source_id():
        return None, f"{os.path.basename(files[-1])} was taken on another build of the kernels (not used)"
    named = [r["mean_per_dispatch"] for r in rows if r["counter"] == "kernel"]
    if kernel is not None and (not named or kernel not in named[0]):
        return None, f"{os.path.basename(files[-1])} profiles {named[0] if named else 'an unnamed kernel'}, this run launched {kernel} (not used)"
    vals = {}
    for r in rows:
        try:
            vals[r["counter"]] = float(r["mean_per_dispatch"])
        except ValueError:
            pass                      # the identity rows (kernel, dispatch ids, kernel_source_id)
    return vals, os.path.basename(files[-1])


def roofline_of(w, local_samples, k_ms, n_flush, kernel_name, with_pmc, instantiation=None):
    """The bench line's `roofline` object for one workload: f64 VALU issue (the real bound), BASELINE's algorithmic-bytes model
    beside it, both from THIS run's kernel time; PMC-derived fields only from a committed profile of this very build."""
    from raytracinginrust_amd import workloads
    per_kind = workloads.F64_OPS_PER_SAMPLE[w.key]
    ops = workloads.valu_ops(per_kind)
    achieved = ops * local_samples / (k_ms * 1e-3) / 1e12
    peak = workloads.F64_VALU_PEAK_OPS / 1e12
    bps = workloads.BYTES_PER_SAMPLE[w.key]
    model = bps * local_samples / (k_ms * 1e-3) / 1e9
    roof = {"bound": "f64_valu", "achieved": achieved, "peak": peak, "unit": "Tops/s (f64 VALU issue-equivalents: lane-operations weighted by issue cost)",
            "frac": achieved / peak,
            "frac_is": "REFERENCE-EQUIVALENT throughput against the issue peak: the reference's algorithmic f64 operations per sample x samples / s.  "
                       "Not the share of issue slots the kernel fills — it executes fewer operations than the reference's arithmetic has (see "
                       "achieved_counts); that share is `executed_valu_frac` (from the PMC counters of the timed frames, null without a profile of this build)",
            "executed_valu_frac": None,
            "peak_measured_issue": workloads.F64_VALU_MEASURED_ISSUE_OPS / 1e12, "frac_of_measured_issue": achieved / (workloads.F64_VALU_MEASURED_ISSUE_OPS / 1e12),
            "traffic": None, "traffic_unit": "bytes per launch (PMC FETCH_SIZE + WRITE_SIZE, KB counters x 1024; raw values)",
            "traffic_source": None, "hbm_measured_GBps": None, "hbm_measured_frac": None,
            "achieved_counts": "the reference's ALGORITHMIC f64 operations per sample x samples / kernel time; since round 5 the kernels execute fewer than that "
                               "(inner BVH boxes by an f32 filter, one exact rect test per Cube instead of six) — what is really issued is `valu_pmc`",
            "ops_per_sample": ops, "flops_per_sample_unweighted": workloads.flops(per_kind),
            "achieved_unweighted_TFLOPs": workloads.flops(per_kind) * local_samples / (k_ms * 1e-3) / 1e12,
            # the same kernel time priced WITHOUT issue weights, against the chip's f64 vector specification (78.6 TFLOP/s counts an FMA
            # as two flops; this path has none): every +, -, x, /, sqrt and libm call counts once.  `frac` above prices the compiler's
            # 11-instruction IEEE divide, the square root and the libm calls at what they cost in issue slots.
            "frac_unweighted": workloads.flops(per_kind) * local_samples / (k_ms * 1e-3) / 78.6e12, "peak_unweighted_TFLOPs": 78.6,
            "divide_share_of_weighted_ops": workloads.VALU_OP_WEIGHTS["div"] * per_kind.get("div", 0.0) / ops,
            "ops_per_sample_source": "raytracinginrust_amd/workloads.py F64_OPS_PER_SAMPLE x VALU_OP_WEIGHTS (the reference's f64 operations by kind, counted by "
                                     "the op-counting build of the CPU oracle on this workload's own grid; tests/sweeps/measure_ops_per_sample.py)",
            "peak_source": "256 CUs x 4 SIMDs x 16 f64 lanes/clk x 2.4 GHz = 39.3e12 lane-operations/s (78.6 TFLOP/s spec counts an FMA as 2; the path has none); "
                           "peak_measured_issue = what a stream of independent f64 adds / multiplies sustains, 64 lanes / 4.92 cycles x 2.26 GHz x 1024 SIMDs "
                           "(profiles/r04_ubench.csv: the weights are cycles relative to that same 4.92)",
            "kernel": kernel_name, "kernel_ms": k_ms, "samples_per_launch": local_samples,
            "framebuffer_atomics_per_launch": 3 * n_flush,       # what the kernel itself counted in THIS run: f64 atomic adds, 8 B each
            "framebuffer_atomic_bytes_per_launch": 24 * n_flush,
            "model_hbm": {"achieved_GBps": model, "peak_GBps": HBM_PEAK_GBPS, "ratio": model / HBM_PEAK_GBPS, "exceeds_hbm_peak": model > HBM_PEAK_GBPS,
                          "bytes_per_sample": bps, "algorithmic_bytes_per_launch": bps * local_samples,
                          "bytes_per_sample_source": "raytracinginrust_amd/workloads.py BYTES_PER_SAMPLE (oracle event counters, tests/sweeps/measure_bytes_per_sample.py)",
                          "note": "BASELINE's metric: algorithmic bytes (event x record-size, SURVEY 8(d)) over the HBM peak.  A MODEL figure, not HBM "
                                  "utilisation: the scene is L2/LDS-resident, physical traffic is `traffic`; a ratio above 1 only says the records "
                                  "never come from HBM"}}
    valu = None
    if with_pmc:
        vals, src = pmc_profile(w.key, instantiation)
        if vals is None:
            roof["traffic_source"] = src
        else:
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                roof["traffic"] = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
                # north_star's literal: HBM GB/s against the chip's peak — the replayed counter bytes over THIS run's kernel time
                roof["hbm_measured_GBps"] = roof["traffic"] / (k_ms * 1e-3) / 1e9
                roof["hbm_measured_frac"] = roof["hbm_measured_GBps"] / HBM_PEAK_GBPS
                roof["traffic_source"] = f"profiles/{src}: committed rocprofv3 --pmc passes of this command on this build (not measured in this run)"
            need = ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_THREAD_CYCLES_VALU")
            waves = vals.get("LAUNCH_WAVES", vals.get("SQ_WAVES"))      # grid size / 64 (SQ_WAVES reports double on some dispatches)
            if waves and all(k in vals for k in need) and vals["SQ_WAVE_CYCLES"] > 0 and vals["SQ_ACTIVE_INST_VALU"] > 0:
                # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles summed over waves; the grid is persistent, so
                # SQ_WAVE_CYCLES / waves is the launch's length in quad-cycles and waves / 1024 the waves per SIMD.
                # (a SIMD that issues vector instructions back to back reads slightly ABOVE 1 here — C2 / C5 1.05 since the room form put f32
                # instructions where f64 divisions were: the counter tallies, per wave, the quad-cycles a vector instruction is in flight, and
                # consecutive instructions of different waves overlap by a pipeline stage.  The line keeps the raw value and uses min(1, raw).)
                busy_raw = vals["SQ_ACTIVE_INST_VALU"] * (waves / N_SIMD) / vals["SQ_WAVE_CYCLES"]
                busy = min(1.0, busy_raw)
                lanes = vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"])
                valu = {"valu_busy_frac": busy, "valu_busy_frac_raw": busy_raw, "valu_lane_utilisation": lanes, "frac": busy * lanes,
                        "frac_meaning": "useful VALU lane-slots (every VALU instruction, not only f64 arithmetic) / lane-slots the SIMDs could have issued over the launch",
                        "source": f"profiles/{src}: committed rocprofv3 --pmc passes of this command on this build (not measured in this run)"}
                roof["executed_valu_frac"] = valu["frac"]
                if "SQ_INSTS_VALU" in vals:
                    valu["valu_wave_instructions_per_sample"] = vals["SQ_INSTS_VALU"] / local_samples
    return roof, valu


FEAT_BITS = ((1, "BVH"), (2, "SPHERES"), (4, "TRIS"), (8, "MEDIUM"), (16, "TEXTURES"), (32, "DIELECTRIC"), (64, "PBR"), (128, "NEAR_FIRST"),
             (256, "PERSIST (persistent traversal)"), (2048, "SPEC (walk-ahead filtered walk)"))


def kernel_name_of(loop):
    """The instantiation that RAN, as the library reports it (rt_last_loop_info) — the name a rocprofv3 trace shows."""
    feats = loop["feats"]
    bits = " | ".join(n for b, n in FEAT_BITS if feats & b) or "lean: rects + instances + Lambertian / Metal / DiffuseLight"
    return f"{loop['kernel']} ({bits}; loop shape {loop['shape']}, chosen by: {loop['chosen_by']})"


def resolve_mode(mode, gpus, env_world):
    """How `python bench.py --gpus N` runs: returns (in-process?, number of GPUs).  Under a launcher (WORLD_SIZE > 1) it is one process
    per GPU and the launcher's world size is the number of GPUs; without one, N > 1 makes this one process drive the N GPUs through
    rt_render_multi_device — never an error, never a re-launch (a driver that runs `python bench.py --gpus 8` bare gets a line).  Only an
    explicit `--mode procs` without a launcher is refused."""
    if mode == "procs" and env_world == 1 and gpus > 1:
        raise ValueError("bench.py --mode procs --gpus N with N > 1 must be launched with torch.distributed.run --nproc-per-node N "
                         "(without a launcher, --mode inproc drives the N GPUs from this one process)")
    inproc = gpus > 1 and env_world == 1 and mode in ("auto", "inproc")
    return inproc, (gpus if inproc else env_world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--also", default="C1,C3,C4,C5",
                    help="other BASELINE configs to time after the main one (one reduced-spp warm-up frame + one full frame each); 'none' to skip")
    ap.add_argument("--mode", default="auto", choices=("auto", "inproc", "procs"),
                    help="N > 1: `procs` = one process per GPU under torch.distributed.run (rt_render_device + one torch.distributed.gather); "
                         "`inproc` = this one process drives the N GPUs through rt_render_multi_device (ncclCommInitAll + one ncclGather); "
                         "auto = procs when WORLD_SIZE is set, inproc otherwise")
    ap.add_argument("--tile-px", type=int, default=67, help="pixels per tile; prime by default (see dist.DEFAULT_TILE_PX)")
    ap.add_argument("--f32", action="store_true", help="throughput variant (not the headline: reduced precision)")
    ap.add_argument("--near-first", action="store_true", help="opt-in RT_NEAR_FIRST_BVH traversal (not the reference's order)")
    ap.add_argument("--pipeline", default="auto", choices=("auto", "1", "2"),
                    help="frames in flight (procs mode).  2 lets a frame's drain overlap the next frame's start on a second stream (measured on one GPU: "
                         "-3 %% per 1/8-frame share).  auto = 1 at N = 1; at N > 1 the headline workload is timed with 1 AND with 2 and the better one "
                         "is reported, both numbers in the line (whether RCCL's gather kernels find room beside a persistent kernel needs N GPUs to tell)")
    ap.add_argument("--sah", action="store_true", help="opt-in RT_BVH_SAH builder (not the reference's tree shape)")
    ap.add_argument("--self-check", default="auto", choices=("auto", "on", "off"),
                    help="after the timed region, render two rows of the frame again on rank 0's device alone and compare them with the step's frame "
                         "(`multi_check`).  auto = at N > 1 (where a wrong un-permute, an idle rank or a stale buffer could hide); at N = 1 it would only add "
                         "two small launches of the same kernel to a profiler's per-kernel averages")
    ap.add_argument("--loop", default="auto", choices=("auto", "persistent", "lockstep"),
                    help="mesh scenes (C4): which loop shape runs — same samples either way.  auto = measured for the view before the warm-up "
                         "(rt_scene_calibrate: four small launches outside every timed region; at N > 1 rank 0's result is handed to every rank); "
                         "the other two set it (rt_scene_set_loop_shape) and skip the measurement — what a profiled run uses so that no "
                         "calibration launch lands in its per-kernel averages.  No effect on the other scenes")
    ap.add_argument("--cpu-spp", type=int, default=-1, help="spp of the bounded CPU-baseline sample (0 = skip, -1 = sized for ~8 s of wall time)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracinginrust_amd import _lib, dist as D, render as R, workloads

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        inproc, args.gpus = resolve_mode(args.mode, args.gpus, env_world)
    except ValueError as e:
        sys.exit(str(e))
    world = 1 if inproc else env_world          # ranks of the process group (the in-process mode has none)
    n_gpus = args.gpus
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
    device_mask, virtual = 0, False
    if inproc:
        n_dev = torch.cuda.device_count()
        if n_dev >= n_gpus:
            device_mask = (1 << n_gpus) - 1
        elif os.environ.get("RT_MULTI_VIRTUAL_RANKS", "") == str(n_gpus):
            device_mask, virtual = 1, True      # test hook: the N ranks' shares one after the other on the one device (csrc/rt_multi.cpp)
        else:
            sys.exit(f"bench.py --gpus {n_gpus}: {n_dev} GPU(s) visible.  (RT_MULTI_VIRTUAL_RANKS={n_gpus} runs the {n_gpus}-rank decomposition on one "
                     "device: a test of the flow, not a measurement.)")

    be = _lib.load()            # after `import torch`: one HIP runtime in the process
    from raytracinginrust_amd import scenes
    earth = None
    flags = (R.RT_F32 if args.f32 else R.RT_F64) | (R.RT_NEAR_FIRST_BVH if args.near_first else 0)
    cdev = "cuda" if backend == "nccl" else "cpu"

    def sync():
        for d in range(torch.cuda.device_count() if inproc else 1):
            torch.cuda.synchronize(d if inproc else None)   # every stream of this process's device(s) (frames run on side streams)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        el = torch.tensor([x], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    if world > 1:
        # communicators and point-to-point channels are created lazily on first use: do that outside every timed region
        tiny = torch.zeros(8, dtype=torch.float64, device=cdev)
        dist.gather(tiny, [torch.empty_like(tiny) for _ in range(world)] if rank == 0 else None, dst=0)
        dist.all_reduce(tiny)

    def build_scene(w):
        nonlocal earth
        if w.scene == "final" and earth is None:
            earth = scenes.load_earthmap()          # the reference's own 1024x512 texture (raytracinginrust_amd/assets/earthmap.jpg)
        b, cam, bg = workloads.build(w, be, earth)
        if args.sah:
            R.set_bvh_builder(b, R.RT_BVH_SAH)
        return b, cam, bg

    def settle_loop_shape(b, cam, bg, w):
        """Before anything is timed: the view-dependent choices the asynchronous entry points never make themselves (rt_scene_calibrate).
        Mesh scenes: fix the loop shape.  The asynchronous entry points the steps use never measure it
        themselves (they would have to wait), so: --loop auto measures it here for this very view — on rank 0's device, and every
        rank then runs what rank 0 found (ranks measuring alone could disagree and the line could not say what ran)."""
        if args.loop != "auto":
            R.set_loop_shape(b, 1 if args.loop == "persistent" else 0)      # (the call below then measures no loop shape; it still tunes a one-BVH world's filter tree)
        # (every rank calls it: for a world that is ONE BVH — C1 — the call tunes the filter tree for the view, host arithmetic that every
        # rank's copy of the scene wants; a mesh scene's four calibration launches run on every rank side by side and rank 0's verdict counts)
        R.calibrate(b, cam, bg, w.W, w.H, w.spp, w.max_depth, flags=flags)
        if world > 1 and args.loop == "auto":
            ch = torch.tensor([R.stored_loop_shape(b) if rank == 0 else -1], dtype=torch.int64, device=cdev)
            dist.broadcast(ch, src=0)
            if rank != 0 and int(ch.item()) >= 0:
                R.set_loop_shape(b, int(ch.item()))

    def check_rows(b, cam, bg, w, frame_rows):
        """Self-validation of a frame, outside every timed region: rows H/2 and H/2 + 1 of the SAME frame rendered again on this device
        alone — tile = one image row, world = H, rank = the row: one launch per row, no sharding, no gather, no un-permute — against
        the rows the step produced (`frame_rows`: 2 x W x 3, host).  Same paths (the RNG is keyed by pixel and sample), so the sums
        agree to summation-order rounding; a wrong un-permute, a rank that rendered nothing or a stale buffer does not."""
        if args.self_check == "off" or (args.self_check == "auto" and n_gpus == 1):
            return None
        r0 = w.H // 2
        ref = []
        for row in (r0, r0 + 1):
            tr = D.TileRenderer(b, cam, bg, w.W, w.H, w.spp, w.max_depth, flags=flags, tile_px=w.W, rank=row, world=w.H)
            ref.append(tr.render_local().clone())
            del tr
        torch.cuda.synchronize()
        ref = torch.stack(ref, 0).reshape(2, w.W, 3).cpu().numpy()
        got = np.asarray(frame_rows, dtype=np.float64).reshape(2, w.W, 3)
        fin = np.isfinite(ref) & np.isfinite(got)
        tol = 1e-12 * (w.spp + np.abs(np.where(fin, ref, 0.0)))
        diff = np.abs(np.where(fin, got, 0.0) - np.where(fin, ref, 0.0))
        ok = bool(np.array_equal(np.isfinite(ref), np.isfinite(got)) and (diff <= tol).all() and float(np.abs(np.where(fin, got, 0.0)).sum()) > 0.0)
        return {"rows": [r0, r0 + 1], "max_abs_diff": float(diff.max()), "tolerance": "1e-12 * (spp + |x|) per channel sum",
                "reference": "the same rows rendered by one launch each on rank 0's device alone (tile = row, no gather, no un-permute)", "ok": ok}

    def run_procs(w, steps, warmup, warm_spp=None, pipeline=1):
        """One process per GPU (or the one GPU): time `steps` frames of workload `w` (after `warmup` untimed ones; `warm_spp` renders the
        warm-up frames at a reduced sample count through the same kernel and scene).  Returns what rank 0 needs for its line."""
        b, cam, bg = build_scene(w)
        # one-time initialisation (scene flatten + upload, code-object load, event creation) is not part of a step
        R.prepare(b, flags)
        settle_loop_shape(b, cam, bg, w)
        if warmup:
            tw = D.TileRenderer(b, cam, bg, w.W, w.H, warm_spp or w.spp, w.max_depth, flags=flags, tile_px=args.tile_px, rank=rank, world=world,
                                pipeline=pipeline)
            for _ in range(warmup):
                tw.render_frame(dst=0)
            sync()
            del tw
        tr = D.TileRenderer(b, cam, bg, w.W, w.H, w.spp, w.max_depth, flags=flags, tile_px=args.tile_px, rank=rank, world=world,
                            pipeline=pipeline)
        sync()
        R.kernel_time_total(b, reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            frame = tr.render_frame(dst=0)             # asynchronous: no host stop between frames
        sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # HIP events around every kernel on its own launch stream, summed by the library (no host stop after each frame)
        k_total_ms, k_launches = R.kernel_time_total(b)
        assert k_launches == steps
        res = {"w": w, "elapsed": elapsed, "steps": steps, "k_ms": k_total_ms / k_launches, "stats": R.last_stats(b), "n_flush": R.last_flush_count(b),
               "pipeline": tr.pipeline, "multi_ms": None, "stats_scope": "rank 0's share" if world > 1 else "frame", "loop": R.last_loop_info(b)}
        n_px = w.W * w.H                     # real (unpadded) pixels this rank's launch owns
        local_px = sum(max(0, min(n_px, (t + 1) * args.tile_px) - t * args.tile_px)
                       for t in D.local_tile_ids(w.W, w.H, args.tile_px, rank, world) if t * args.tile_px < n_px)
        res["local_samples"] = local_px * w.spp
        # what every rank did, as seen by the collective itself: [rank, device, kernel ms, samples] all-gathered over the process group
        mine = torch.tensor([float(rank), float(dev), res["k_ms"], float(res["local_samples"])], dtype=torch.float64, device=cdev)
        rows = [mine.clone() for _ in range(world)]
        if world > 1:
            dist.all_gather(rows, mine)
        res["ranks"] = {"ranks_seen": dist.get_world_size() if world > 1 else 1, "devices": [int(r[1].item()) for r in rows],
                        "kernel_ms": [float(r[2].item()) for r in rows], "local_samples": [int(r[3].item()) for r in rows],
                        "hosts_rank_ids": [int(r[0].item()) for r in rows]}
        if rank == 0:
            assert frame is not None and tuple(frame.shape) == (w.H, w.W, 3)
            res["mean_radiance"] = float(torch.nan_to_num(frame).mean().item()) / w.spp
            res["multi_check"] = check_rows(b, cam, bg, w, frame[w.H // 2: w.H // 2 + 2].cpu().numpy())
        del tr, frame, b
        torch.cuda.empty_cache()
        return res

    def run_inproc(w, steps, warmup, warm_spp=None, pipeline=1):
        """This one process, N GPUs: every step is one rt_render_multi_device call — N persistent launches on N streams, one ncclGather,
        the un-permute on the first device — with the frame left in that device's memory (as the N = 1 bench leaves it); the calls do
        not wait for each other's frames beyond what the library's one-frame-in-flight rule asks for."""
        b, cam, bg = build_scene(w)
        R.prepare(b, flags)
        settle_loop_shape(b, cam, bg, w)        # on this process's current device; the stored shape serves every device's launch of this view
        for _ in range(max(1, warmup)):         # (the first call also creates the communicators and uploads the scene everywhere)
            R.render_multi_device(b, cam, bg, w.W, w.H, (warm_spp or w.spp) if warmup else 1, w.max_depth, device_mask, flags=flags, tile_px=args.tile_px)
        R.multi_sync(b)
        sync()
        R.kernel_time_total(b, reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            R.render_multi_device(b, cam, bg, w.W, w.H, w.spp, w.max_depth, device_mask, flags=flags, tile_px=args.tile_px)
        R.multi_sync(b)
        sync()
        elapsed = time.perf_counter() - t0
        k_total_ms, k_launches = R.kernel_time_total(b)
        assert k_launches == steps * n_gpus
        frame = R.multi_frame(b, w.W, w.H)
        rk = R.last_multi_ranks(b)             # the last frame's ranks as the library ran them: device, kernel ms; ranks in RCCL's communicator
        n_px = w.W * w.H
        tile_px = args.tile_px or D.DEFAULT_TILE_PX
        per_rank = [sum(max(0, min(n_px, (t + 1) * tile_px) - t * tile_px) for t in D.local_tile_ids(w.W, w.H, tile_px, r, n_gpus) if t * tile_px < n_px) * w.spp
                    for r in range(n_gpus)]
        res = {"w": w, "elapsed": elapsed, "steps": steps, "k_ms": k_total_ms / k_launches, "stats": R.last_stats(b), "n_flush": R.last_flush_count(b),
               "pipeline": 1, "multi_ms": R.last_multi_ms(b), "mean_radiance": float(np.nan_to_num(frame).mean()) / w.spp,
               "local_samples": w.samples / n_gpus,            # per launch: k_ms is the mean over the frame's N launches, so is this
               "stats_scope": "frame",                         # rt_last_stats sums the counters of the frame's N launches
               "loop": R.last_loop_info(b),
               "ranks": {"ranks_seen": rk["collective_ranks"] if rk["collective_ranks"] else rk["n_ranks"],
                         "ranks_seen_source": "ncclCommCount of the gather's communicator" if rk["collective_ranks"] else
                                              "launches of the frame (no collective ran: virtual ranks on one device)",
                         "devices": rk["devices"], "kernel_ms": rk["kernel_ms"], "local_samples": per_rank}}
        torch.cuda.set_device(rk["devices"][0] if rk["devices"] else 0)
        res["multi_check"] = check_rows(b, cam, bg, w, frame[w.H // 2: w.H // 2 + 2])
        del b
        torch.cuda.empty_cache()
        return res

    import numpy as np
    run = run_inproc if inproc else run_procs
    w = workloads.WORKLOADS[args.workload]
    pipelines = [1] if (inproc or n_gpus == 1 and args.pipeline == "auto") else ([1, 2] if args.pipeline == "auto" else [int(args.pipeline)])
    tried = {pl: run(w, args.steps, args.warmup, pipeline=pl) for pl in pipelines}
    best = min(tried, key=lambda pl: tried[pl]["elapsed"])         # (max-over-ranks times: the same choice on every rank)
    main_res = tried[best]
    extra = []
    if args.also.lower() != "none":
        for key in args.also.split(","):
            if key and key != w.key:
                we = workloads.WORKLOADS[key]
                # one full frame each; a frame of a few milliseconds (C1) is timed over 20 so that the figure is not launch jitter
                extra.append(run(we, 20 if we.samples < 50_000_000 else 1, 1, warm_spp=max(1, we.spp // 32), pipeline=best))

    if rank == 0:
        with_pmc = n_gpus == 1 and not args.f32 and not args.near_first and not args.sah
        roof, valu = roofline_of(w, main_res["local_samples"], main_res["k_ms"], main_res["n_flush"], kernel_name_of(main_res["loop"]), with_pmc,
                                 main_res["loop"]["kernel"])
        cpu = None
        if n_gpus == 1 and args.cpu_spp != 0 and not args.f32:
            cpu = cpu_baseline(w, args.cpu_spp, earth)
        st = main_res["stats"]
        others = {}
        for r in extra:
            we = r["w"]
            ro, va = roofline_of(we, r["local_samples"], r["k_ms"], r["n_flush"], kernel_name_of(r["loop"]), with_pmc, r["loop"]["kernel"])
            others[we.key] = {"workload": we.describe(), "value": we.samples * r["steps"] / r["elapsed"] / 1e6, "unit": "Msamples/s", "steps": r["steps"],
                              "warmup": f"1 frame at {max(1, we.spp // 32)} spp (same kernel and scene)", "ms_per_step": r["elapsed"] / r["steps"] * 1e3,
                              "kernel_ms": r["k_ms"], "frac": ro["frac"], "frac_unweighted": ro["frac_unweighted"], "frac_of_measured_issue": ro["frac_of_measured_issue"],
                              "achieved_unweighted_TFLOPs": ro["achieved_unweighted_TFLOPs"], "ops_per_sample": ro["ops_per_sample"], "achieved_Tops": ro["achieved"],
                              "model_hbm_ratio": ro["model_hbm"]["ratio"], "exceeds_hbm_peak": ro["model_hbm"]["exceeds_hbm_peak"],
                              "hbm_measured_GBps": ro["hbm_measured_GBps"], "hbm_measured_frac": ro["hbm_measured_frac"],
                              "bytes_per_sample": ro["model_hbm"]["bytes_per_sample"],
                              "model_hbm_GBps": ro["model_hbm"]["achieved_GBps"], "traffic": ro["traffic"], "traffic_source": ro["traffic_source"],
                              "valu_pmc": va, "executed_valu_frac": ro["executed_valu_frac"], "kernel": ro["kernel"], "loop": r["loop"], "multi_ms": r["multi_ms"],
                              "lane_utilisation": r["stats"]["live_lane_iterations"] / max(1, 64 * r["stats"]["wave_iterations"]),
                              "nonfinite_samples": r["stats"]["nonfinite_samples"], "nonfinite_samples_scope": r["stats_scope"],
                              "mean_radiance": r["mean_radiance"], "multi_check": r.get("multi_check"), "ranks": r.get("ranks")}
        if inproc:
            par = (f"one process, rt_render_multi_device: {n_gpus} " + ("VIRTUAL ranks on one device (RT_MULTI_VIRTUAL_RANKS test hook: the decomposition, "
                   "not a measurement; device-to-device copies stand in for the gather)" if virtual else "GPUs, ncclCommInitAll + 1 ncclGather + un-permute on device 0"))
        else:
            par = f"one process per GPU: tiles interleaved over {n_gpus} GPU(s) + 1 torch.distributed.gather ({backend})"
        out = {
            "metric": "Msamples/s (pixels x spp / s)", "value": w.samples * args.steps / main_res["elapsed"] / 1e6, "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64", "data": "synthetic",
            "config": {"workload": f"{w.key}: {w.describe()}", "tile_px": args.tile_px, "seed": "0x5EED",
                       "parallelism": par, "mode": "inproc" if inproc else ("procs" if n_gpus > 1 else "single"),
                       "frames_in_flight": main_res["pipeline"]},
            "roofline": roof, "valu_pmc": valu, "loop": main_res["loop"], "cpu_baseline": cpu, "workloads": others,
            "lane_utilisation": st["live_lane_iterations"] / max(1, 64 * st["wave_iterations"]),
            "nonfinite_samples": st["nonfinite_samples"], "nonfinite_samples_scope": main_res["stats_scope"], "mean_radiance": main_res["mean_radiance"],
        }
        if inproc:
            out["config"]["frames_consumed"] = ("no: every step's frame stays in device 0's memory (two buffers, alternating), only the last one is read "
                                                "back after the timed region — as the N = 1 bench leaves its frames in HBM")
        if len(tried) > 1:
            out["pipeline_tried"] = {str(pl): {"ms_per_step": r["elapsed"] / args.steps * 1e3, "value": w.samples * args.steps / r["elapsed"] / 1e6}
                                     for pl, r in tried.items()}
        if main_res["multi_ms"] is not None:
            out["multi_ms_last_frame"] = main_res["multi_ms"]
        # self-validation (outside the timed region): two rows of the timed frame against a plain one-device render of those rows, and
        # what each rank did.  `multi_ok` is false if any workload's check failed, a rank rendered nothing, or fewer ranks than asked ran.
        out["multi_check"] = main_res.get("multi_check")
        out["ranks"] = main_res.get("ranks")
        checks = [main_res.get("multi_check")] + [r.get("multi_check") for r in extra]
        rk = main_res.get("ranks") or {}
        out["multi_ok"] = bool(all(c is None or c["ok"] for c in checks) and (n_gpus == 1 or all(c is not None for c in checks)) and rk.get("ranks_seen") == n_gpus and len(rk.get("kernel_ms", [])) == n_gpus
                               and all(k > 0.0 for k in rk.get("kernel_ms", [])) and all(x > 0 for x in rk.get("local_samples", []))
                               and (virtual or len(set(rk.get("devices", []))) == n_gpus))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
