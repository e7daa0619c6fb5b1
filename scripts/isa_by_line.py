#!/usr/bin/env python3
"""Instruction mix of one kernel per source line, from a -gline-tables-only -save-temps assembly:
python scripts/isa_by_line.py file.s mangled_prefix [min_count]
Columns: VALU (incl. DPP / lane ops), MFMA, SALU, LDS, VMEM, s_nop cycles, s_waitcnt."""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
files, cur, infn = {}, None, False
cnt = collections.defaultdict(collections.Counter)
for l in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if l.startswith(fn):
        infn = True
    if infn and l.strip().startswith('.Lfunc_end'):
        infn = False
    if not infn:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    op = t.split()[0]
    c = cnt[cur]
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif op.startswith('v_'): c['valu'] += 1; c[op] += 1
    elif op == 's_nop': c['nop'] += int(t.split()[1]) + 1
    elif op == 's_waitcnt': c['wait'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c['vmem'] += 1
tot = collections.Counter()
rows = []
for loc, c in cnt.items():
    for k in ('valu', 'mfma', 'salu', 'lds', 'vmem', 'nop', 'wait'):
        tot[k] += c[k]
    rows.append((loc, c))
print("total", dict(tot))
rows.sort(key=lambda x: -(x[1]['valu'] + x[1]['nop'] // 4))
for loc, c in rows:
    if c['valu'] + c['mfma'] + c['lds'] + c['nop'] < minc:
        continue
    top = ", ".join("%s %d" % (k, v) for k, v in c.most_common(12) if k.startswith('v_'))
    print("%-16s %4d  valu %4d mfma %3d salu %3d lds %3d vmem %3d nop %4d wait %3d | %s" % (files.get(loc[0], '?') if loc else '?', loc[1] if loc else 0, c['valu'], c['mfma'], c['salu'], c['lds'], c['vmem'], c['nop'], c['wait'], top[:150]))
