#!/usr/bin/env python3
"""Where does a kernel spill?  Count scratch loads/stores per source line of a -S -gline-tables-only
assembly: python scripts/spills_by_line.py /tmp/a2.s _Z10k_fwd_bwd2ILi128"""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
cur, infn = None, False
files = {}
cnt = collections.Counter()
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
    if t.startswith('scratch_store') or t.startswith('scratch_load'):
        w = {'dword': 1, 'dwordx2': 2, 'dwordx3': 3, 'dwordx4': 4}[t.split()[0].split('_')[-1]]
        cnt[(cur, t.split('_')[1])] += w
for (loc, op), n in sorted(cnt.items(), key=lambda x: (x[0][0] or (0, 0), x[0][1])):
    print("%-22s line %4d  %-5s %3d dwords" % (files.get(loc[0], loc[0]) if loc else "?", loc[1] if loc else 0, op, n))
