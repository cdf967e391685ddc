"""The rows of DESIGN.md §3.1's measured table from the committed profile set of a round.   usage: python tools/roofline_table.py r05"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
notes = {"C1": " (lock-step, walk-ahead filtered walk, 1024-thread workgroups)", "C4": " (persistent traversal, 768-thread workgroups)"}
for W in ("C1", "C2", "C3", "C4", "C5"):
    d = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{W}.json")).read())
    r = d["roofline"]
    pm = {x["counter"]: x["mean_per_dispatch"] for x in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{W}_pmc_summary.csv")))}
    f = lambda k: float(pm[k])
    # the instantiation is the one the summary was taken on AND the bench line says it launched (round 6: both name it)
    kern = pm.get("kernel") or d["roofline"]["kernel"]
    assert "loop" not in d or d["loop"]["kernel"] == kern, (W, d["loop"]["kernel"], kern)
    names = {W: "`" + kern.replace("rt::", "") + "`" + notes.get(W, "")}
    busy = min(1.0, f("SQ_ACTIVE_INST_VALU") * (f("LAUNCH_WAVES") / 1024) / f("SQ_WAVE_CYCLES"))      # (raw 1.05 on C2 / C5: bench.py says why)
    lanes = f("SQ_THREAD_CYCLES_VALU") / (64 * f("SQ_ACTIVE_INST_VALU"))
    fetch, write = f("FETCH_SIZE") * 1024 / 1e9, f("WRITE_SIZE") * 1024 / 1e9
    gbps = (fetch + write) / (r["kernel_ms"] * 1e-3)
    ks = [x for x in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{W}_kernel_stats.csv"))) if "pathtrace" in x["Name"]][0]
    print(f"| {W} | {names[W]} | {r['kernel_ms']:.1f} (rocprofv3: {float(ks['AverageNs']) / 1e6:.1f} avg, {float(ks['MaxNs']) / 1e6:.1f} max over {ks['Calls']} launches) | {d['value']:.0f} | {r['achieved']:.1f} | "
          f"**{r['frac']:.3f}** ({r['frac_of_measured_issue']:.2f}); unweighted {r['frac_unweighted']:.3f} | {busy:.2f} × {lanes:.2f} = {busy * lanes:.2f} | {fetch:.3f} + {write:.2f} GB | "
          f"{gbps:.1f} GB/s = {gbps / 8000 * 100:.2f} % | {r['model_hbm']['ratio']:.2f}{' (flagged)' if r['model_hbm']['exceeds_hbm_peak'] else ''} |")
    print(f"    VALU wave-instructions per sample {f('SQ_INSTS_VALU') / r['samples_per_launch']:.1f}, kernel_source_id {pm['kernel_source_id']}", file=sys.stderr)
