"""Summarise rocprofv3 --pmc CSVs (tools/profile_pmc.sh: one rocprofv3 run per counter set under <root>/<pass>/) into <root>/summary.csv:
per counter, the median over the TIMED FRAMES of the bench command — and only those.

Which dispatches are the timed frames (round 6; until then: every path-tracing dispatch with a large grid, median — which for a mesh
scene picked a loop-shape calibration launch of the other instantiation):
  * the kernel name must contain the instantiation the bench line says it launched (`--kernel`, or `loop.kernel` of the JSON line at the
    end of <root>/<pass>.log: e.g. `rt::pathtrace_kernel<double, 261u>`);
  * among those, the launch geometry (Grid_Size, Workgroup_Size) of the longest dispatch, and a duration (End - Start timestamp) of at
    least half the longest's: warm-up frames at reduced sample counts, calibration launches (<= 1024 x 1024 x 16 samples) and the
    self-check's single rows fall out; a full-size warm-up frame is the same work as a timed one and stays.
The summary names the kernel, lists the dispatch ids it used per pass and how many path-tracing dispatches it dropped, and the run FAILS
(exit status 1, no summary written) when the kept dispatches disagree by more than 1 % in a counter that does not depend on timing
(instruction and thread counts).  `LAUNCH_WAVES` = Grid_Size / 64 (SQ_WAVES itself reports twice the waves on some dispatches).

    python3 tools/pmc_summary.py <root> [--kernel 'pathtrace_kernel<double, 261u>'] [--tolerance 0.01]
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

STABLE = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")   # work done, not time taken


def kernel_from_log(root, pass_name):
    """`loop.kernel` of the bench line the pass printed (the last line of <root>/<pass>.log that parses as JSON)."""
    try:
        lines = open(os.path.join(root, pass_name + ".log")).read().splitlines()
    except OSError:
        return None
    for line in reversed(lines):
        if line.startswith("{"):
            try:
                return json.loads(line)["loop"]["kernel"]
            except (ValueError, KeyError, TypeError):
                return None
    return None


def timed_frames(rows, kernel):
    """(ids of the dispatches that are timed frames, number of other path-tracing dispatches) among one pass's rows."""
    by = collections.OrderedDict()
    others = set()
    for r in rows:
        if "pathtrace" not in r["Kernel_Name"]:
            continue
        if kernel.replace("rt::", "") in r["Kernel_Name"]:
            by.setdefault(int(r["Dispatch_Id"]), r)
        else:
            others.add(int(r["Dispatch_Id"]))
    if not by:
        return [], len(others)
    dur = {d: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for d, r in by.items()}
    longest = max(dur, key=lambda d: dur[d])
    geom = (by[longest]["Grid_Size"], by[longest]["Workgroup_Size"])
    keep = [d for d, r in by.items() if (r["Grid_Size"], r["Workgroup_Size"]) == geom and 2 * dur[d] >= dur[longest]]
    return keep, len(others) + len(by) - len(keep)


def summarise(root, kernel=None, tolerance=0.01):
    """Returns (lines of the summary CSV, list of problems)."""
    res = collections.OrderedDict()
    used, problems, dropped = [], [], 0
    kernels = set()
    files = sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")))
    if not files:
        return [], [f"no counter_collection.csv under {root}"]
    for f in files:
        pass_name = os.path.relpath(f, root).split(os.sep)[0]
        k = kernel or kernel_from_log(root, pass_name)
        if not k:
            problems.append(f"pass {pass_name}: no --kernel given and no bench line with loop.kernel in {pass_name}.log")
            continue
        kernels.add(k)
        rows = list(csv.DictReader(open(f)))
        keep, n_drop = timed_frames(rows, k)
        dropped += n_drop
        if not keep:
            problems.append(f"pass {pass_name}: no dispatch of {k}")
            continue
        used.append(f"{pass_name}:" + " ".join(str(d) for d in keep))
        for r in rows:
            if int(r["Dispatch_Id"]) in keep and k.replace("rt::", "") in r["Kernel_Name"]:
                res.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                    res.setdefault("LAUNCH_WAVES", []).append(float(int(r["Grid_Size"]) // 64))
    if len(kernels) > 1:
        problems.append(f"the passes launched different instantiations: {sorted(kernels)}")
    out = []
    for name, v in res.items():
        vs = sorted(v)
        mean = vs[len(vs) // 2] if len(vs) % 2 else 0.5 * (vs[len(vs) // 2 - 1] + vs[len(vs) // 2])      # the MEDIAN of the timed frames: the first one
        # of a process reads the scene and the code object from memory (FETCH_SIZE 48x the others' on the teapot room), and SQ_WAVES reports
        # twice the waves on some dispatches; the column keeps its historical name
        if name in STABLE and mean > 0 and (max(v) - min(v)) > tolerance * mean:
            problems.append(f"{name}: the {len(v)} timed frames disagree by {(max(v) - min(v)) / mean:.1%} (min {min(v):.6g}, max {max(v):.6g})")
        out.append(f"{name},{mean:.6g},{len(v)},{min(v):.6g},{max(v):.6g}")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from raytracinginrust_amd import buildinfo
    out.append(f"kernel,\"{sorted(kernels)[0] if kernels else ''}\",{len(used)},,")
    out.append(f"timed_dispatches,\"{'; '.join(used)}\",{len(used)},,")
    out.append(f"dropped_pathtrace_dispatches,{dropped},0,,")      # warm-ups at reduced spp, calibration launches, self-check rows
    out.append(f"kernel_source_id,{buildinfo.kernel_source_id()},0,,")      # bench.py quotes these counters only on the same kernels
    return out, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--tolerance", type=float, default=0.01)
    a = ap.parse_args()
    out, problems = summarise(a.root, a.kernel, a.tolerance)
    if problems:
        print("pmc_summary: NOT written:\n  " + "\n  ".join(problems), file=sys.stderr)
        return 1
    open(os.path.join(a.root, "summary.csv"), "w").write("counter,mean_per_dispatch,dispatches,min,max\n" + "\n".join(out) + "\n")
    print("\n".join(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
