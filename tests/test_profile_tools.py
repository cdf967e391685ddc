"""tools/pmc_summary.py picks the TIMED FRAMES of the instantiation the bench line names out of a rocprofv3 --pmc CSV — not warm-ups at
reduced sample counts, not the loop-shape calibration launches (which for a mesh scene include dispatches of the OTHER instantiation
with the same large grid: round 5's summary of C4 was the median of seven dispatches, i.e. a lock-step calibration launch)."""
import csv
import importlib.util
import json
import os

from conftest import ROOT

spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
pmc_summary = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pmc_summary)

COLS = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size",
        "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
PERS = "void rt::pathtrace_kernel<double, 261u>(rt::KParams<double>)"
LOCK = "void rt::pathtrace_kernel<double, 5u>(rt::KParams<double>)"


def _write_pass(root, name, dispatches, counters, line=None):
    """dispatches: (id, kernel, grid, wg, duration_ns, {counter: value})"""
    d = os.path.join(root, name, "host")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(COLS)
        t = 1000
        for did, kern, grid, wg, dur, vals in dispatches:
            for c in counters:
                w.writerow([did, did, "Agent 2", 1, 1, 1, grid, 8, kern, wg, 0, 0, 128, 0, 96, c, float(vals[c]), t, t + dur])
            t += dur + 10
    if line is not None:
        open(os.path.join(root, name + ".log"), "w").write("some banner\n" + json.dumps(line) + "\n")


def _c4_like(root, third=7.5207e11):
    # bench --steps 2 --warmup 1 on the teapot room as round 5 ran it: four calibration launches (two per loop shape, the lock-step
    # ones with the LARGER grid), then three full frames
    disp = [(13, "__amd_rocclr_fillBufferAligned", 65536, 256, 900, {"SQ_INSTS_VALU": 7e4, "SQ_WAVE_CYCLES": 1e6}),
            (14, LOCK, 262144, 1024, 5_000_000, {"SQ_INSTS_VALU": 2.43e9, "SQ_WAVE_CYCLES": 1.14e10}),
            (18, PERS, 196608, 768, 4_600_000, {"SQ_INSTS_VALU": 1.88e9, "SQ_WAVE_CYCLES": 8.1e9}),
            (22, LOCK, 262144, 1024, 5_000_000, {"SQ_INSTS_VALU": 2.43e9, "SQ_WAVE_CYCLES": 1.14e10}),
            (26, PERS, 196608, 768, 4_600_000, {"SQ_INSTS_VALU": 1.88e9, "SQ_WAVE_CYCLES": 8.1e9}),
            (30, PERS, 196608, 768, 1_590_000_000, {"SQ_INSTS_VALU": 7.5207e11, "SQ_WAVE_CYCLES": 2.9129e12}),
            (34, PERS, 196608, 768, 1_591_000_000, {"SQ_INSTS_VALU": 7.5207e11, "SQ_WAVE_CYCLES": 2.9128e12}),
            (38, PERS, 196608, 768, 1_592_000_000, {"SQ_INSTS_VALU": third, "SQ_WAVE_CYCLES": 2.9130e12})]
    line = {"metric": "Msamples/s", "loop": {"shape": "persistent", "feats": 261, "kernel": "rt::pathtrace_kernel<double, 261u>"}}
    _write_pass(root, "sq1", disp, ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES"], line)
    _write_pass(root, "fetch", [(d[0], d[1], d[2], d[3], d[4], {"FETCH_SIZE": 1876.0 if d[4] > 1e9 else 3.0}) for d in disp], ["FETCH_SIZE"], line)


def test_summary_is_the_timed_frames_of_the_named_instantiation(tmp_path):
    root = str(tmp_path)
    _c4_like(root)
    out, problems = pmc_summary.summarise(root)
    assert problems == []
    rows = {r[0]: r for r in csv.reader(out)}
    assert abs(float(rows["SQ_INSTS_VALU"][1]) - 7.5207e11) < 1e7 and rows["SQ_INSTS_VALU"][2] == "3"       # not 2.43e9, the old median
    assert abs(float(rows["FETCH_SIZE"][1]) - 1876.0) < 1e-9
    assert float(rows["LAUNCH_WAVES"][1]) == 196608 // 64
    assert rows["kernel"][1] == "rt::pathtrace_kernel<double, 261u>"
    assert rows["timed_dispatches"][1] == "fetch:30 34 38; sq1:30 34 38"
    assert int(rows["dropped_pathtrace_dispatches"][1]) == 8                  # four calibration launches in each of the two passes
    assert "kernel_source_id" in rows
    # an explicit --kernel overrides the log: the lock-step calibration launches are then all there is of that instantiation
    out, problems = pmc_summary.summarise(root, kernel="rt::pathtrace_kernel<double, 5u>")
    assert problems == [] and abs(float({r[0]: r for r in csv.reader(out)}["SQ_INSTS_VALU"][1]) - 2.43e9) < 1.0


def test_summary_fails_when_the_timed_frames_disagree_or_nothing_matches(tmp_path):
    root = str(tmp_path / "a")
    _c4_like(root, third=7.0e11)                                              # the third frame did 7 % less work: not the same frame
    out, problems = pmc_summary.summarise(root)
    assert any("SQ_INSTS_VALU" in p and "disagree" in p for p in problems)
    assert pmc_summary.summarise(root, tolerance=0.2)[1] == []
    root2 = str(tmp_path / "b")
    _c4_like(root2)
    out, problems = pmc_summary.summarise(root2, kernel="rt::pathtrace_kernel<double, 63u>")
    assert any("no dispatch" in p for p in problems)
    os.remove(os.path.join(root2, "sq1.log"))
    assert any("no --kernel given" in p for p in pmc_summary.summarise(root2)[1])
    assert pmc_summary.summarise(str(tmp_path / "empty"))[1]


def test_bench_refuses_a_profile_of_another_instantiation(tmp_path, monkeypatch):
    """bench.py replays PMC counters only from a summary taken on this build AND naming the instantiation the run launched."""
    import bench
    from raytracinginrust_amd import buildinfo
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    f = prof / "r99_bench_C4_pmc_summary.csv"
    body = ("counter,mean_per_dispatch,dispatches,min,max\nSQ_INSTS_VALU,7.5207e+11,3,7.5e11,7.6e11\nFETCH_SIZE,1876,3,1,2\nWRITE_SIZE,3.3e6,3,1,2\n"
            "kernel,\"rt::pathtrace_kernel<double, 261u>\",5,,\ntimed_dispatches,\"sq1:30 34 38\",1,,\nkernel_source_id,%s,0,,\n")
    f.write_text(body % buildinfo.kernel_source_id())
    vals, src = bench.pmc_profile("C4", "rt::pathtrace_kernel<double, 261u>")
    assert vals is not None and vals["SQ_INSTS_VALU"] == 7.5207e11 and src == f.name and "kernel" not in vals
    vals, why = bench.pmc_profile("C4", "rt::pathtrace_kernel<double, 5u>")
    assert vals is None and "261u" in why and "5u" in why
    f.write_text(body % "0123456789abcdef")
    vals, why = bench.pmc_profile("C4", "rt::pathtrace_kernel<double, 261u>")
    assert vals is None and "another build" in why
