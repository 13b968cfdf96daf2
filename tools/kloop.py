"""container: memory / wait / spill instructions of one kernel's assembly, with line indices (to see where a loop waits
and what it spills).  usage: python tools/kloop.py /tmp/mfs_visc.s <mangled-name-prefix> [from_barrier_index to_barrier_index]"""
import sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
start = next(k for k, l in enumerate(L) if l.startswith(pre) and ':' in l)
end = next(k for k in range(start, len(L)) if L[k].startswith('.Lfunc_end'))
lines = [l.strip() for l in L[start:end] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
idx = [n for n, l in enumerate(lines) if l.startswith('s_barrier')]
print('instructions', len(lines), 'barriers at', idx)
def cat(l):
    op = l.split()[0]
    for p, c in (('scratch_load', 'sload'), ('scratch_store', 'sstore'), ('v_accvgpr', 'acc'), ('v_fma_f64', 'f64'), ('v_mul_f64', 'f64'),
                 ('v_add_f64', 'f64'), ('v_fmac_f64', 'f64'), ('v_cvt', 'cvt'), ('global_load', 'gload'), ('global_store', 'gstore'),
                 ('ds_', 'ds'), ('v_readlane', 'lane'), ('v_writelane', 'lane'), ('v_mov', 'vmov'), ('s_waitcnt', 'wait')):
        if op.startswith(p): return c
    return 'valu' if op.startswith('v_') else ('salu' if op.startswith('s_') else 'other')
prev = 0
for b in idx + [len(lines)]:
    print(prev, b, dict(Counter(cat(l) for l in lines[prev:b])))
    prev = b
if len(sys.argv) > 4:
    a, b = idx[int(sys.argv[3])], idx[int(sys.argv[4])]
    for n in range(a, b + 1):
        l = lines[n]
        if l.split()[0].startswith(('s_waitcnt', 'global_', 'ds_', 's_barrier', 'scratch', 's_cbranch', 's_branch')) or l.endswith(':'):
            print(n, l[:120])
