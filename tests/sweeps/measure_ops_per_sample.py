"""f64 operations per sample of every BASELINE workload, by kind, from the op-counting build of the CPU oracle (oracle/orc_opcount.h:
`double` replaced by a counting stand-in, same samples) on the workload's OWN pixel grid under the default seed.  The per-kind means
go into raytracinginrust_amd/workloads.py (F64_OPS_PER_SAMPLE); bench.py prices them against the f64 VALU issue peak at every N and
never re-measures them.  Uses the oracle, so it lives under tests/.   usage: python tests/sweeps/measure_ops_per_sample.py [C1 C2 ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from raytracinginrust_amd import scenes, workloads  # noqa: E402

SPP = {"C1": 16, "C2": 4, "C3": 4, "C4": 2, "C5": 1}

if __name__ == "__main__":
    be = orc.load_opcount()
    earth = scenes.load_earthmap()
    out = {}
    for key in (sys.argv[1:] or list(workloads.WORKLOADS)):
        w = workloads.WORKLOADS[key]
        b, cam, bg = workloads.build(w, be, earth)
        orc.op_counts(be)                                  # scene construction is not part of a sample
        spp = SPP[key]
        t = time.perf_counter()
        orc.render(b, cam, bg, w.W, w.H, spp, w.max_depth)
        dt = time.perf_counter() - t
        n = w.W * w.H * spp
        per = {k: round(v / n, 2) for k, v in orc.op_counts(be).items()}
        out[key] = per
        print(f"{key}: {w.describe()}; measured on {w.W}x{w.H} at {spp} spp = {n / 1e6:.1f} M samples, {dt:.1f} s -> "
              f"{workloads.valu_ops(per):.0f} f64 VALU issue-equivalents per sample", flush=True)
    print(json.dumps(out))
