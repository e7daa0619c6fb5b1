#!/usr/bin/env python3
"""Per kernel name of a rocprofv3 --kernel-trace run: median duration and median idle time of the queue before it, and the
name of its predecessor: python scripts/micro/boundary_report.py <dir>"""
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f)))
by = {}
for i in range(1, len(ev)):
    s, e, n = ev[i]
    key = (n, ev[i - 1][2])
    by.setdefault(key, []).append(((e - s) / 1e3, (s - ev[i - 1][1]) / 1e3))
for (n, prev), v in sorted(by.items()):
    if len(v) < 5: continue
    print("%-40s after %-40s dur %7.2f us  gap %6.2f us  (n=%d)" % (n, prev, st.median(x[0] for x in v), st.median(x[1] for x in v), len(v)))
