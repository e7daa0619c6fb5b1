#!/usr/bin/env python3
"""Instruction sequence of the d=128 fp32 training kernel between its stamps, from a -DTLSAN_STAMPS=1 -save-temps
assembly: python scripts/isa_segs.py file.s [min_lines]   (L load, S store, A atomic, M MFMA, d LDS, e exp, . VALU, nK s_nop K,
[..] waitcnt, | branch, # barrier)"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
fn = sys.argv[3] if len(sys.argv) > 3 else '_Z9k_fwd_bwdILi128ELi16ELb1ELb0ELi0ELb0ELi0EEv7FwdArgs:'
a = next(i for i, l in enumerate(src) if l.startswith(fn))
b = next(i for i in range(a, len(src)) if 's_endpgm' in src[i])
lines = src[a:b]
marks = [i for i, l in enumerate(lines) if 's_memtime' in l]
def seq(a, b):
    out = []
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith((';', '.')): continue
        if t.endswith(':'): out.append('\n' + t); continue
        op = t.split()[0]
        if op.startswith(('global_load', 'scratch_load')): out.append('L')
        elif op.startswith('global_atomic'): out.append('A')
        elif op.startswith(('global_store', 'scratch_store')): out.append('S')
        elif op.startswith('s_waitcnt'): out.append('[' + t.split(None, 1)[1].replace('vmcnt', 'v').replace('lgkmcnt', 'k').replace(' ', '') + ']')
        elif op.startswith('v_mfma'): out.append('M')
        elif op.startswith('ds_'): out.append('d')
        elif op.startswith('s_cbranch'): out.append('|')
        elif op.startswith('s_barrier'): out.append('#')
        elif op.startswith('s_nop'): out.append('n' + t.split()[1])
        elif op.startswith('v_exp'): out.append('e')
        elif op.startswith('v_'): out.append('.')
    s = ''.join(out)
    return re.sub(r'\.{4,}', lambda m: '.%d.' % len(m.group()), s)
print(len(lines), marks)
for k in range(len(marks) - 1):
    print('--- seg', k, 'lines', marks[k + 1] - marks[k])
    if marks[k + 1] - marks[k] > minl: print(seq(marks[k], marks[k + 1]))
