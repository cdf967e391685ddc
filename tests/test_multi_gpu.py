"""rt_render_multi (csrc/rt_multi.cpp): the whole frame on several GPUs of one node from ONE C call — what a host that owns the
node's GPUs itself (the reference's `main`, src/main.rs:767-835) calls.  A GPU box of this pool has ONE device, so:
  * N = 1 through the same entry point, with and without the forced RCCL gather (ncclCommInitAll + ncclGather on one rank);
  * the N-rank tile arithmetic and the on-device un-permute kernel through the test hook RT_MULTI_VIRTUAL_RANKS (N ranks that all
    live on the one device);
  * on a node with more GPUs the real N-device path runs (skipped here otherwise).
The CPU part checks the argument handling and that the entry points exist."""
import os

import numpy as np
import pytest

from raytracinginrust_amd import render as R, scenes


def test_render_multi_without_gpu_fails_loudly(pbe):
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    b, cam, bg = scenes.cornell_box(pbe)
    with pytest.raises(R.RenderError, match="no HIP device"):
        R.render_multi(b, cam, bg, 16, 16, 2, 5)


@pytest.mark.gpu
def test_render_multi_one_device_equals_render(pbe):
    b, cam, bg = scenes.cornell_box(pbe)
    W, H, spp, depth = 101, 67, 16, 50
    ref = R.render(b, cam, bg, W, H, spp, depth)
    for flags, tile in ((0, 0), (R.RT_MULTI_COLLECTIVE, 0), (R.RT_MULTI_COLLECTIVE, 64), (0, 7)):
        got = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1, flags=flags, tile_px=tile)
        assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref))), (flags, tile)       # same samples; only the sum order differs
    ms = R.last_multi_ms(b)
    assert ms["slowest_kernel_ms"] > 0 and ms["call_ms"] >= ms["slowest_kernel_ms"]
    with pytest.raises(R.RenderError, match="not visible"):
        R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1 << R.device_count())


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,tile", [(2, 67), (3, 64), (8, 67), (8, 5)])
def test_render_multi_tile_arithmetic_with_virtual_ranks(pbe, ranks, tile, monkeypatch):
    """N ranks on one device: shares rendered one after the other, device-to-device "gather", the real un-permute kernel."""
    b, cam, bg = scenes.cornell_box(pbe, aspect_ratio=16 / 9)
    W, H, spp, depth = 160, 90, 8, 50
    ref = R.render(b, cam, bg, W, H, spp, depth)
    monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", str(ranks))
    got = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1, tile_px=tile)
    assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref)))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 4])
def test_render_multi_device_leaves_the_frame_on_the_device(pbe, ranks, monkeypatch):
    """rt_render_multi_device / rt_multi_sync / rt_multi_copy_frame: the frame stays in the first device's memory (what bench.py's
    in-process N > 1 mode times), consecutive calls reuse the scene's buffers, timings settle at the sync."""
    import torch
    b, cam, bg = scenes.cornell_box(pbe, aspect_ratio=16 / 9)
    W, H, spp, depth = 160, 90, 8, 50
    ref = R.render(b, cam, bg, W, H, spp, depth)
    if ranks > 1:
        monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", str(ranks))
    dev_before = torch.cuda.current_device()
    ptrs = set()
    for _ in range(3):                                   # back-to-back frames without a host wait in between
        ptrs.add(R.render_multi_device(b, cam, bg, W, H, spp, depth, device_mask=1))
    R.multi_sync(b)
    # consecutive frames alternate between two frame buffers on the first device (frame i stays readable while frame i + 1 is un-permuted)
    assert len(ptrs) == 2 and 0 not in ptrs and torch.cuda.current_device() == dev_before
    got = R.multi_frame(b, W, H)
    assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref)))
    ms = R.last_multi_ms(b)
    assert ms["slowest_kernel_ms"] > 0 and ms["call_ms"] > 0 and ms["unpermute_ms"] > 0
    n = R.kernel_time_total(b)[1]
    assert n == 1 + 3 * ranks                            # every rank's launch is timed


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_render_multi_counters_cover_the_whole_frame(pbe, ranks, monkeypatch):
    """rt_last_stats after an N-rank frame reports the frame, not one rank's share: the non-finite samples (here a scene that poisons
    hundreds of samples, main.rs:97 with pdf 0) and the accumulator flushes of the N launches summed equal the single launch's count /
    cover every pixel.  (8 virtual ranks on one device reuse the stream's two launch slots: the counters of the earlier shares must not
    be lost when their slot is taken again.)"""
    from raytracinginrust_amd.api import Camera, Plane, SceneBuilder
    b = SceneBuilder(pbe)
    white = b.Lambertian(b.ConstantTexture((0.73, 0.73, 0.73)))
    floor = b.AARect(Plane.XZ, -100.0, 100.0, -100.0, 100.0, 0.0, white)
    cube = b.Cube((-10.0, 0.0, -10.0), (10.0, 20.0, 10.0), white)           # Cube has no pdf_value / random of its own (hit.rs:29-30)
    world = b.HittableList()
    world.push(floor); world.push(cube)
    b.set_scene(world, [cube])
    cam = Camera((0.0, 50.0, -120.0), (0.0, 5.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    bg = (0.5, 0.7, 1.0)
    W, H, spp, depth = 96, 54, 8, 10
    ref = R.render(b, cam, bg, W, H, spp, depth)
    n_bad = R.last_stats(b)["nonfinite_samples"]
    assert n_bad > 100
    monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", str(ranks))
    got = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1, tile_px=67)
    assert R.last_stats(b)["nonfinite_samples"] == n_bad
    assert R.last_flush_count(b) >= W * H                 # every pixel was handed in at least once, over all shares
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin) and np.all(np.abs(got[fin] - ref[fin]) <= 1e-12 * (spp + np.abs(ref[fin])))
    monkeypatch.delenv("RT_MULTI_VIRTUAL_RANKS")
    R.render(b, cam, bg, W, H, spp, depth)                # a later single launch is a frame of its own again
    assert R.last_stats(b)["nonfinite_samples"] == n_bad


@pytest.mark.gpu
def test_render_multi_failure_leaves_nothing_behind(pbe, monkeypatch):
    """A rank that fails after others were launched (test hook RT_MULTI_FAIL_RANK): the call returns an error with the rank in the
    message, nothing stays in flight, no temporary leaks (device memory in use does not grow over repeated failures), the caller's
    current device is untouched, and the next call renders the right frame."""
    import torch
    b, cam, bg = scenes.cornell_box(pbe, aspect_ratio=16 / 9)
    W, H, spp, depth = 160, 90, 8, 50
    ref = R.render(b, cam, bg, W, H, spp, depth)
    monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", "4")
    good = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1)            # allocates the scene's multi buffers once
    torch.cuda.synchronize()
    dev_before = torch.cuda.current_device()
    free0 = torch.cuda.mem_get_info()[0]
    monkeypatch.setenv("RT_MULTI_FAIL_RANK", "2")
    for _ in range(30):
        with pytest.raises(R.RenderError, match="rank 2.*injected failure"):
            R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1)
        with pytest.raises(R.RenderError, match="injected failure"):
            R.render_multi_device(b, cam, bg, W, H, spp, depth, device_mask=1)
    R.multi_sync(b)                                      # nothing pending: returns at once
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == dev_before
    # (sixty failing calls: one leaked tile buffer per call would be tens of megabytes; the HIP runtime's own pools grow in 2 MiB steps at
    # moments of their choosing — one such step inside this window failed the test once at a tolerance of 1 MiB)
    assert torch.cuda.mem_get_info()[0] >= free0 - (4 << 20)
    monkeypatch.delenv("RT_MULTI_FAIL_RANK")
    again = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1)
    assert np.array_equal(again, good) and np.all(np.abs(again - ref) <= 1e-12 * (spp + np.abs(ref)))


@pytest.mark.gpu
def test_last_multi_ranks_reports_every_rank(pbe, monkeypatch):
    """rt_last_multi_ranks: per rank of the last rt_render_multi* frame the HIP device and the kernel time (also when 8 virtual ranks
    reuse a stream's two launch slots), and the size RCCL itself reports for the communicator the gather ran on."""
    b, cam, bg = scenes.cornell_box(pbe)
    W, H, spp, depth = 160, 120, 16, 20
    R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1, flags=R.RT_MULTI_COLLECTIVE)
    rk = R.last_multi_ranks(b)
    assert rk["n_ranks"] == 1 and rk["devices"] == [0] and rk["collective_ranks"] == 1 and rk["kernel_ms"][0] > 0.0
    R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1)
    assert R.last_multi_ranks(b)["collective_ranks"] == 0                  # one device, no collective asked for
    for n in (3, 8):
        monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", str(n))
        R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=1)
        rk = R.last_multi_ranks(b)
        assert rk["n_ranks"] == n and rk["devices"] == [0] * n and rk["collective_ranks"] == 0
        assert all(k > 0.0 for k in rk["kernel_ms"])
    monkeypatch.delenv("RT_MULTI_VIRTUAL_RANKS")


@pytest.mark.gpu
def test_loop_calibration_stays_out_of_a_multi_frames_counters(pbe, monkeypatch):
    """A large mesh frame measures its loop shape with four small launches at the scene's first render (rt_host.cpp: calibrate_loop_shape).
    Inside an rt_render_multi frame — whose N launches share one frame number so that rt_last_stats can sum them — those launches must
    not be counted: the frame's counters equal a plain single-launch render's."""
    from test_parity_gpu import _cloud_room
    W, H, spp, depth = 1024, 1024, 100, 20
    b, cam, bg = _cloud_room(pbe, 400, 70.0)
    ref = R.render(b, cam, bg, W, H, spp, depth)
    want = R.last_stats(b)
    b2, cam2, bg2 = _cloud_room(pbe, 400, 70.0)                    # a fresh scene: its first render is the multi frame
    monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", "2")
    got = R.render_multi(b2, cam2, bg2, W, H, spp, depth, device_mask=1)
    monkeypatch.delenv("RT_MULTI_VIRTUAL_RANKS")
    st = R.last_stats(b2)
    assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref)))
    assert st["live_lane_iterations"] == want["live_lane_iterations"] and st["nonfinite_samples"] == want["nonfinite_samples"]


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_runs_in_process_without_a_launcher(gpus):
    """`python bench.py --gpus N` with no launcher (what the driver's SCALE run may use): the in-process mode through
    rt_render_multi_device, here with N virtual ranks on the one GPU; the line keeps the contract."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RT_MULTI_VIRTUAL_RANKS=str(gpus))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--also", "C1",
                        "--cpu-spp", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "inproc"
    assert "rt_render_multi_device" in d["config"]["parallelism"] and "VIRTUAL" in d["config"]["parallelism"]
    assert abs(d["value"] - 800 * 800 * 1024 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "f64_valu" and 0 < d["roofline"]["frac"] <= 1 and d["cpu_baseline"] is None
    assert set(d["multi_ms_last_frame"]) == {"slowest_kernel_ms", "gather_ms", "unpermute_ms", "call_ms"}
    assert 0.13 < d["mean_radiance"] < 0.18 and "C1" in d["workloads"]          # the Cornell frame's mean per channel (oracle: 0.155)
    # round 5: the line validates itself — two rows of the timed frame against a plain one-device render of those rows, every rank's
    # device / kernel time / share, the number of ranks that took part (here: launches, no collective runs between virtual ranks)
    mc, rk = d["multi_check"], d["ranks"]
    assert mc["ok"] and mc["rows"] == [400, 401] and mc["max_abs_diff"] <= 1e-12 * (1024 + 1024 * 15.0)
    assert d["workloads"]["C1"]["multi_check"]["ok"] and d["workloads"]["C1"]["ranks"]["ranks_seen"] == gpus
    assert rk["ranks_seen"] == gpus and len(rk["devices"]) == gpus and len(rk["kernel_ms"]) == gpus and all(k > 0 for k in rk["kernel_ms"])
    assert sum(rk["local_samples"]) == 800 * 800 * 1024 and max(rk["local_samples"]) / min(rk["local_samples"]) < 1.01
    assert d["multi_ok"] is True


@pytest.mark.gpu
def test_bench_runs_one_process_per_gpu_under_the_launcher():
    """The driver's documented N > 1 launch — torch.distributed.run, one rank per GPU — here with 2 ranks sharing the one GPU over gloo
    (RT_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device): rank 0 prints one contract-valid line, `--pipeline auto` times the
    headline with one and with two frames in flight and reports the better one with both numbers."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("RT_MULTI_VIRTUAL_RANKS", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--also", "C1,C4"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "procs" and "torch.distributed.gather" in d["config"]["parallelism"]
    assert set(d["pipeline_tried"]) == {"1", "2"} and d["config"]["frames_in_flight"] in (1, 2)
    assert d["ms_per_step"] == min(v["ms_per_step"] for v in d["pipeline_tried"].values())
    assert abs(d["value"] - 800 * 800 * 1024 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None and d["roofline"]["traffic"] is None and 0.13 < d["mean_radiance"] < 0.18
    mc, rk = d["multi_check"], d["ranks"]
    assert mc["ok"] and mc["rows"] == [400, 401]
    assert rk["ranks_seen"] == 2 and rk["hosts_rank_ids"] == [0, 1] and all(k > 0 for k in rk["kernel_ms"]) and sum(rk["local_samples"]) == 800 * 800 * 1024
    assert d["workloads"]["C1"]["multi_check"]["ok"]
    # round 6: the mesh scene's loop shape is measured by rank 0 before anything is timed and handed to the other rank (rt_scene_calibrate,
    # a broadcast, rt_scene_set_loop_shape): the line says which instantiation ran and why, and both ranks' shares assemble to the frame
    c4 = d["workloads"]["C4"]
    assert c4["multi_check"]["ok"] and c4["loop"]["chosen_by"] == "calibration of this view" and c4["loop"]["calibration_ms"]["persistent"] > 0
    assert c4["loop"]["kernel"] in ("rt::pathtrace_kernel<double, 261u>", "rt::pathtrace_kernel<double, 5u>") and c4["loop"]["kernel"] in c4["kernel"]
    assert d["loop"]["shape"] == "list" and d["loop"]["kernel"] == "rt::pathtrace_kernel<double, 0u>"
    # (two ranks share the one GPU here, so `multi_ok` — which asks for N distinct devices — is false by design on this box)
    assert d["multi_ok"] is False and rk["devices"] == [0, 0]


@pytest.mark.gpu
def test_render_multi_random_frame_shapes_and_tile_sizes(pbe, monkeypatch):
    """The N-rank decomposition over random frame shapes, tile sizes (also larger than a row, larger than the frame, 1) and rank counts:
    rt_render_multi (virtual ranks on the one device, the real un-permute kernel) == rt_render, and so is the one-process-per-GPU
    decomposition (rt_render_device per rank + dist.assemble) for the same shape."""
    import torch
    from raytracinginrust_amd import dist as D
    b, cam, bg = scenes.cornell_box(pbe, aspect_ratio=1.3)
    rs = np.random.RandomState(7)
    for case in range(24):
        W, H = int(rs.randint(2, 97)), int(rs.randint(2, 61))
        ranks = int(rs.choice([2, 3, 5, 7, 8, 16]))
        tile = int(rs.choice([1, 2, 7, 64, 67, 100, W, W + 1, W * H, W * H + 5]))
        spp = int(rs.randint(1, 5))
        ref = R.render(b, cam, bg, W, H, spp, 12)
        monkeypatch.setenv("RT_MULTI_VIRTUAL_RANKS", str(ranks))
        got = R.render_multi(b, cam, bg, W, H, spp, 12, device_mask=1, tile_px=tile)
        monkeypatch.delenv("RT_MULTI_VIRTUAL_RANKS")
        assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref))), (W, H, ranks, tile, spp)
        parts = []
        for rank in range(ranks):
            parts.append(D.TileRenderer(b, cam, bg, W, H, spp, 12, tile_px=tile, rank=rank, world=ranks).render_local().clone())
        torch.cuda.synchronize()
        frame = D.assemble(torch.stack(parts), W, H, tile).cpu().numpy()
        assert np.all(np.abs(frame - ref) <= 1e-12 * (spp + np.abs(ref))), (W, H, ranks, tile, spp)


@pytest.mark.gpu
def test_render_multi_all_devices(pbe):
    n = R.device_count()
    if n < 2:
        pytest.skip("one GPU on this box: the N-device path is covered by the virtual-rank test and the N = 1 collective")
    b, cam, bg = scenes.cornell_box(pbe)
    W, H, spp, depth = 400, 400, 64, 50
    ref = R.render(b, cam, bg, W, H, spp, depth)
    got = R.render_multi(b, cam, bg, W, H, spp, depth, device_mask=0)
    assert np.all(np.abs(got - ref) <= 1e-12 * (spp + np.abs(ref)))


@pytest.mark.gpu
def test_cxx_host_gpus_flag(pbe):
    """`rtrender --gpus 1 --collective`: the C++ host through rt_render_multi prints the image `rtrender` prints through rt_render."""
    import subprocess
    from raytracinginrust_amd import _lib
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "host", "rtrender")
    args = ["--scene", "cornell", "--width", "48", "--height", "27", "--spp", "8", "--depth", "20"]
    a = subprocess.run([exe] + args, check=True, capture_output=True, text=True).stdout
    c = subprocess.run([exe] + args + ["--gpus", "1", "--collective"], check=True, capture_output=True, text=True)
    assert "rt_render_multi:" in c.stderr
    la, lc = a.split("\n"), c.stdout.split("\n")
    assert la[:3] == lc[:3] and len(la) == len(lc)
    diff = sum(1 for x, y in zip(la[3:], lc[3:]) if x != y)
    assert diff <= 3                                   # 8-bit quantisation ties under a different summation order
