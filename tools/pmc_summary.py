"""Summarise rocprofv3 --pmc CSVs: per counter, the MEDIAN over the dispatches of the path-tracing kernel (the first dispatch of a
pass sometimes reports a doubled SQ_WAVES; the column keeps its historical name mean_per_dispatch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv'))):
    for r in csv.DictReader(open(f)):
        if 'pathtrace' not in r['Kernel_Name']: continue
        if int(r.get('Grid_Size', r.get('Grid_Size_X', '1000000')) or 1000000) < 65536: continue      # ignore tiny auxiliary launches
        res.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        # the launch's wave count from its geometry (SQ_WAVES itself reports twice the waves on some dispatches)
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': res.setdefault('LAUNCH_WAVES', []).append(float(int(r['Grid_Size']) // 64))
out = []
for k, v in res.items():
    vs = sorted(v)
    med = vs[len(vs) // 2] if len(vs) % 2 else 0.5 * (vs[len(vs) // 2 - 1] + vs[len(vs) // 2])
    out.append(f"{k},{med:.6g},{len(v)}")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracinginrust_amd import buildinfo
out.append(f"kernel_source_id,{buildinfo.kernel_source_id()},0")      # bench.py quotes these counters only on the same kernels
open(os.path.join(root, 'summary.csv'), 'w').write("counter,mean_per_dispatch,dispatches\n" + "\n".join(out) + "\n")
print("\n".join(out))
