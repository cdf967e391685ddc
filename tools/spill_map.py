"""Where a kernel's register-spill code sits, by innermost enclosing loop, with what each loop is (round 6: the question "which values are
spilled INSIDE the box-step loop").  Reads the kernel's body from a `hipcc --cuda-device-only -S` file (LLVM annotates every block with
its loop: 'in Loop: Header=BBn_m Depth=d' / 'This [Inner] Loop Header: Depth=d' / 'Parent Loop BBn_m Depth=d').
usage: spill_map.py file.s [mangled kernel name]        (without a name: the whole file as one body)
Per innermost loop: instructions, scratch loads / stores, LDS reads (ds_read_b128 = a filter-node fetch: the loops with two per iteration
are the box steps), f64 divisions' v_div_fmas (a primitive test), and the header chain."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
if len(sys.argv) > 2:
    m = re.search(r'^' + re.escape(sys.argv[2]) + r':[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, flags=re.S | re.M)
    txt = m.group(0)
lines = txt.split('\n')
header_of = {}      # block -> (innermost header, depth)
parents = {}        # header -> parent header
cur = ('entry', 0)
stats = collections.defaultdict(lambda: collections.Counter())
label = None
for l in lines:
    m = re.match(r'^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)\s*(;.*)?$', l)
    if m:
        label = m.group(1)
        a = m.group(2) or ''
        mi = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', a)
        mh = re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)', a)
        if mh and label:
            cur = (label, int(mh.group(1)))
        elif mi:
            cur = (mi.group(1), int(mi.group(2)))
        elif not a.strip():
            cur = ('(no loop)', 0)
        continue
    mp = re.match(r'^;\s+(?:Parent Loop|Child Loop) (BB\d+_\d+) Depth=(\d+)', l)
    if mp:
        if 'Parent Loop' in l and cur[0] not in ('entry', '(no loop)'):
            # the chain is printed outermost first: remember the last parent seen with depth = cur depth - 1
            if int(mp.group(2)) == cur[1] - 1:
                parents[cur[0]] = mp.group(1)
        continue
    st = l.strip()
    if not st or st.startswith(';') or st.startswith('.'):
        continue
    c = stats[cur]
    c['insts'] += 1
    op = st.split()[0]
    if op.startswith('scratch_load'): c['sload'] += 1
    if op.startswith('scratch_store'): c['sstore'] += 1
    if op == 'ds_read_b128' or op == 'ds_load_b128': c['lds128'] += 1
    if op == 'v_div_fmas_f64': c['div'] += 1
    if op.startswith('v_readlane') or op.startswith('v_writelane'): c['sgpr_spill'] += 1
    if op.startswith('global_load') or op.startswith('s_load'): c['gload'] += 1


def chain(h):
    out = [h]
    while out[-1] in parents:
        out.append(parents[out[-1]])
    return ' < '.join(out)


print(f"{'innermost loop (header @ depth)':34s} {'insts':>6s} {'s_load':>6s} {'s_store':>7s} {'ds128':>5s} {'f64div':>6s} {'v_r/wlane':>9s}  enclosing")
tot = collections.Counter()
for (h, d), c in sorted(stats.items(), key=lambda kv: (-kv[0][1], -kv[1]['insts'])):
    for k, v in c.items(): tot[(d, k)] += v
    if c['sload'] or c['sstore'] or d >= 3 or c['insts'] > 300:
        print(f"{h + ' @' + str(d):34s} {c['insts']:6d} {c['sload']:6d} {c['sstore']:7d} {c['lds128']:5d} {c['div']:6d} {c['sgpr_spill']:9d}  {chain(h)}")
print('--- by depth: instructions, scratch loads, scratch stores')
for d in sorted({k[0] for k in tot}):
    print(f"depth {d}: insts {tot[(d, 'insts')]:6d}  scratch_load {tot[(d, 'sload')]:4d}  scratch_store {tot[(d, 'sstore')]:4d}  v_read/writelane {tot[(d, 'sgpr_spill')]:4d}")
