"""The op-counting build of the CPU oracle (oracle/orc_opcount.h: `double` replaced by a counting stand-in) renders the SAME samples
as the normal build, and its tallies are what raytracinginrust_amd/workloads.py commits as F64_OPS_PER_SAMPLE (the f64-VALU roofline's
algorithmic operation counts)."""
import numpy as np
import pytest

from conftest import build_scene
from raytracinginrust_amd import workloads


@pytest.fixture(scope="module")
def ops_be():
    from oracle import orc
    return orc.load_opcount()


@pytest.mark.parametrize("name,W,H,spp,depth", [("cornell", 24, 24, 8, 50), ("random", 32, 18, 4, 8), ("final", 16, 16, 4, 50), ("teapot", 32, 18, 4, 50)])
def test_counting_build_renders_the_same_samples(name, W, H, spp, depth, obe, ops_be, earth):
    from oracle import orc
    b0, cam0, bg0 = build_scene(name, obe, earth)
    b1, cam1, bg1 = build_scene(name, ops_be, earth)
    ref, rs = orc.render(b0, cam0, bg0, W, H, spp, depth, want_samples=True)
    orc.op_counts(ops_be)
    got, gs = orc.render(b1, cam1, bg1, W, H, spp, depth, want_samples=True)
    assert np.array_equal(gs.view(np.uint64), rs.view(np.uint64)) and np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    n = orc.op_counts(ops_be)
    assert n["add"] > 0 and n["mul"] > 0 and n["div"] > 0 and n["sqrt"] > 0
    assert all(v == 0 for v in orc.op_counts(ops_be).values())          # reset on read
    assert all(v == 0 for v in orc.op_counts(obe).values())             # the normal build tallies nothing


def test_committed_op_counts_match_a_fresh_count(ops_be):
    """C2's committed per-kind table (counted at 4 spp) against a fresh count of the same frame at 1 spp: the means agree within
    Monte-Carlo error."""
    from oracle import orc
    w = workloads.WORKLOADS["C2"]
    b, cam, bg = workloads.build(w, ops_be)
    orc.op_counts(ops_be)
    orc.render(b, cam, bg, w.W, w.H, 1, w.max_depth)
    n_samples = w.W * w.H
    per = {k: v / n_samples for k, v in orc.op_counts(ops_be).items()}
    want = workloads.F64_OPS_PER_SAMPLE["C2"]
    assert workloads.valu_ops(per) == pytest.approx(workloads.valu_ops(want), rel=0.02)
    assert per["div"] == pytest.approx(want["div"], rel=0.02) and per["sqrt"] == pytest.approx(want["sqrt"], rel=0.02)
    assert set(workloads.F64_OPS_PER_SAMPLE) == set(workloads.WORKLOADS) == set(workloads.BYTES_PER_SAMPLE)
    for key, tab in workloads.F64_OPS_PER_SAMPLE.items():
        assert set(tab) <= set(workloads.VALU_OP_WEIGHTS) and workloads.valu_ops(tab) > workloads.flops(tab) > 100
