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
