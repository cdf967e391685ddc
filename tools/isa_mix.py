"""Instruction mix of each kernel in a hipcc -save-temps .s file (static counts)."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/rt_kernel-hip-amdgcn-amd-amdhsa-gfx950.s'
lines = open(path).read().split('\n')
cur = None; kern = collections.OrderedDict()
for l in lines:
    m = re.match(r'^(_ZN2rt[A-Za-z0-9_]+):', l)
    if m: cur = m.group(1); kern[cur] = []; continue
    if cur is None: continue
    st = l.strip()
    if st.startswith('s_endpgm'): kern[cur].append('s_endpgm'); cur = None; continue
    if l.startswith('\t') and st and not st.startswith(('.', ';')): kern[cur].append(st.split()[0])
for name, ins in kern.items():
    c = collections.Counter(ins); tot = len(ins)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(name, 'total', tot)
    print('  f64 VALU', g(lambda k: '_f64' in k), '| div_scale', c['v_div_scale_f64'], 'div_fmas', c['v_div_fmas_f64'], 'div_fixup', c['v_div_fixup_f64'],
          'rcp', g(lambda k: k.startswith('v_rcp_f64')), 'rsq', g(lambda k: k.startswith('v_rsq_f64')), 'sqrt', g(lambda k: k.startswith('v_sqrt_f64')),
          'fma', c['v_fma_f64'], 'mul', g(lambda k: k.startswith('v_mul_f64')), 'add', g(lambda k: k.startswith('v_add_f64')), 'cmp', g(lambda k: k.startswith('v_cmp') and 'f64' in k))
    print('  f32 VALU', g(lambda k: '_f32' in k))
    print('  s_load', g(lambda k: k.startswith('s_load')), 'global_load', g(lambda k: k.startswith('global_load')), 'global_store', g(lambda k: k.startswith('global_store')),
          'atomic', g(lambda k: 'atomic' in k), 'scratch', g(lambda k: k.startswith('scratch_')), 'ds', g(lambda k: k.startswith('ds_')))
    print('  v_cndmask', g(lambda k: k.startswith('v_cndmask')), 's_waitcnt', c['s_waitcnt'], 'branches', g(lambda k: k.startswith('s_cbranch') or k == 's_branch'),
          'v_mov', g(lambda k: k.startswith('v_mov')), 'SALU', g(lambda k: k.startswith('s_') and not k.startswith(('s_load', 's_waitcnt', 's_cbranch', 's_branch', 's_nop'))), 's_nop', c['s_nop'])
