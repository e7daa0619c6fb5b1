#!/usr/bin/env python3
"""Where does a kernel drain its vector-memory queue behind a spill reload?  A scratch_load that is followed within a few
instructions by `s_waitcnt vmcnt(0)` waits for EVERYTHING in flight (loads, stores and returning atomics retire in order),
so a reloaded address or operand in the middle of a gather costs a full round trip.  Counts them per source line of a
-S -gline-tables-only assembly (hipcc --cuda-device-only -S -gline-tables-only unit.hip -o unit.s):
  python scripts/drains_by_line.py unit.s <mangled kernel name prefix> [top]"""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
infn, cur, files, seq = False, None, {}, []
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
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    seq.append((cur, t))
c = collections.Counter()
for i, (loc, t) in enumerate(seq):
    if t.startswith('scratch_load'):
        for j in range(i + 1, min(i + 6, len(seq))):
            if seq[j][1].startswith('s_waitcnt') and 'vmcnt(0)' in seq[j][1]:
                k = j
                while k < len(seq) and (seq[k][0] is None or seq[k][0][1] == 0):
                    k += 1
                c[seq[k][0] if k < len(seq) else loc] += 1
                break
print("instructions %d, scratch loads %d, scratch stores %d, reloads followed by vmcnt(0) %d, vmcnt(0) waits in all %d" % (
    len(seq), sum(1 for _, t in seq if t.startswith('scratch_load')), sum(1 for _, t in seq if t.startswith('scratch_store')),
    sum(c.values()), sum(1 for _, t in seq if t.startswith('s_waitcnt') and 'vmcnt(0)' in t)))
for k, v in c.most_common(top):
    print("%-24s line %5d  %3d" % (k[0], k[1], v))
