"""Estimated dynamic instruction mix of the step kernel from its ISA (hipcc -S): static counts per region weighted by
trip counts (substep body x4, PGS sweep x32).   usage: tools/isa_mix.py /tmp/k.s sub_start sub_end pgs_start pgs_end"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
sub_start, sub_end, pgs_start, pgs_end = [int(a) for a in sys.argv[2:6]]
def classify(op):
    if op.startswith(('v_fma', 'v_fmac', 'v_fmamk', 'v_fmaak', 'v_pk_fma')): return 'fma'
    if op.startswith(('v_mul_f32', 'v_pk_mul')): return 'mul'
    if op.startswith(('v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_pk_add')): return 'add'
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if op.startswith('v_mov'): return 'mov'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_xor'): return 'xor'
    if op.startswith(('v_max', 'v_min')): return 'minmax'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_div')): return 'div/trans'
    if op.startswith('v_'): return 'v_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    return 'other'
def count(a, b):
    c = collections.Counter()
    for l in lines[a:b]:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if m and not l.strip().startswith(('.', ';')): c[classify(m.group(1))] += 1
    return c
body = count(sub_start, sub_end); pgs = count(pgs_start, pgs_end)
dyn = collections.Counter()
for k, v in body.items(): dyn[k] += 4 * (v - pgs.get(k, 0))
for k, v in pgs.items(): dyn[k] += 32 * v
for reg in (count(0, sub_start), count(sub_end, len(lines))):
    for k, v in reg.items(): dyn[k] += v
tot = sum(v for k, v in dyn.items() if k not in ('lds', 'waitcnt', 'salu', 'vmem', 'other'))
print("estimated dynamic VALU instructions per step:", tot)
for k, v in dyn.most_common(): print("%-10s %7d %5.1f%%" % (k, v, 100 * v / tot))
print("static: substep body", sum(body.values()), " PGS sweep", sum(pgs.values()))
