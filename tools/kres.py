"""Resource usage (VGPRs, scratch, spills, occupancy, code size) of every kernel in (a patched copy of) rt_kernel.hip — compile only.
usage: python tools/kres.py [file.hip] [--tu 2] [--flags "-DRT_KRES_ONLY=63u ..."]"""
import argparse, os, re, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("src", nargs="?", default="rt_kernel.hip"); ap.add_argument("--tu", default="2"); ap.add_argument("--keep", default=None)
ap.add_argument("--flags", default="")
a = ap.parse_args()
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracinginrust_amd", "csrc")
extra = a.flags.split()
if a.tu == "1": extra = ["-mllvm", "-enable-misched=0"] + extra
obj = a.keep or f"/tmp/kres_{os.getpid()}.o"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "--offload-arch=gfx950",
       f"-DRT_TU={a.tu}", "-I" + csrc, "-c", a.src if os.path.isabs(a.src) else os.path.join(csrc, a.src), "-o", obj, "-Rpass-analysis=kernel-resource-usage"] + extra
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-3000:])
rows, cur = [], None
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for k, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                   ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)")):
        m = re.search(pat, l)
        if m and cur is not None: cur[k] = int(m.group(1))
for x in rows:
    m = re.search(r"kernelI(\w)Lj(\d+)E", x["name"])
    tag = (m.group(1) + m.group(2)) if m else x["name"][:48]
    print(f"{tag:>10}  vgpr {x.get('vgpr'):>4}  scratch {x.get('scratch'):>5}  vspill {x.get('vspill'):>4}  sspill {x.get('sspill'):>4}  occ {x.get('occ')}")
if not a.keep: os.remove(obj)
