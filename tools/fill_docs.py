"""Fill the @Cn..@ placeholders of README.md / BASELINE.md / DESIGN.md from the committed default bench line + per-workload profiles of a round.
usage: python tools/fill_docs.py r05"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
tag = sys.argv[1]
d = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench_default.json")).read().strip().splitlines()[-1])
rows = {"C2": {"value": d["value"], "ms": d["ms_per_step"], "frac": d["roofline"]["frac"], "fm": d["roofline"]["frac_of_measured_issue"], "fu": d["roofline"]["frac_unweighted"],
               "model": d["roofline"]["model_hbm"]["ratio"], "flag": d["roofline"]["model_hbm"]["exceeds_hbm_peak"], "k_ms": d["roofline"]["kernel_ms"]}}
for k, e in d["workloads"].items():
    rows[k] = {"value": e["value"], "ms": e["ms_per_step"], "frac": e["frac"], "fm": e["frac_of_measured_issue"], "fu": e["frac_unweighted"], "model": e["model_hbm_ratio"],
               "flag": e["exceeds_hbm_peak"], "k_ms": e["kernel_ms"]}
sub = {}
for k, r in rows.items():
    pm = {x["counter"]: x["mean_per_dispatch"] for x in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{k}_pmc_summary.csv")))}
    gb = (float(pm["FETCH_SIZE"]) + float(pm["WRITE_SIZE"])) * 1024 / 1e9
    gbps = gb / (r["k_ms"] * 1e-3)
    v = r["value"]
    sub[f"@{k}v@"] = f"{v:,.0f}".replace(",", " ") if v >= 10000 else f"{v:.0f}"
    sub[f"@{k}ms@"] = f"{r['ms']:.1f}" if r["ms"] < 1000 else f"{r['ms'] / 1e3:.2f}"
    sub[f"@{k}f@"] = f"{r['frac']:.3f}"; sub[f"@{k}fm@"] = f"{r['fm']:.2f}"; sub[f"@{k}fu@"] = f"{r['fu']:.3f}"
    sub[f"@{k}h@"] = f"{gbps:.1f} GB/s" + (f" = {gbps / 80:.2f} % of peak" if k == "C2" else "")
    sub[f"@{k}m@"] = f"{r['model']:.2f}" + (" (flagged)" if r["flag"] else "")
for name in sys.argv[2:]:
    p = os.path.join(ROOT, name); s = open(p).read()
    for a, b in sub.items(): s = s.replace(a, b)
    left = re.findall(r"@C\d\w+@", s)
    assert not left, left
    open(p, "w").write(s)
print(json.dumps(sub, indent=0))
