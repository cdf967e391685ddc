"""Where the spill code of a kernel sits: for every scratch_load / scratch_store of a -save-temps .s file, the loop nest of its block
(LLVM prints 'Parent Loop ... Depth=N' / 'This Inner Loop Header: Depth=N' after block labels).  usage: scratch_map.py file.s"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
depth = 0; label = None; hist = collections.Counter(); per_block = collections.OrderedDict()
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        label = m.group(1); d = 0; j = i
        # the label line and the following comment lines carry the loop annotations
        while j < len(lines) and (j == i or lines[j].lstrip().startswith(';')):
            for mm in re.finditer(r'Depth=(\d+)', lines[j]): d = max(d, int(mm.group(1)))
            j += 1
        depth = d
    st = l.strip()
    if st.startswith('scratch_'):
        kind = 'load' if 'load' in st else 'store'
        hist[(depth, kind)] += 1
        per_block.setdefault((label, depth), [0, 0])[0 if kind == 'load' else 1] += 1
    i += 1
for (d, k), n in sorted(hist.items()): print(f'loop depth {d}: {n:4d} scratch_{k}')
print('blocks with spill code at depth >= 3:')
for (lab, d), (nl, ns) in per_block.items():
    if d >= 3: print(f'  {lab:14s} depth {d}: {nl} loads, {ns} stores')
