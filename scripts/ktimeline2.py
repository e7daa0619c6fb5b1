#!/usr/bin/env python3
"""Steady-state kernel timeline of a rocprofv3 --kernel-trace run: python scripts/ktimeline2.py <dir> [first] [count]
prints start / end / duration of consecutive dispatches and the idle time of the main queue before each of its kernels."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 800
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:30], r['Queue_Id']) for r in rows)
mainq = next(q for s, e, n, q in ev if 'k_fwd_bwd' in n)
t0 = ev[first][0]
last_end = {}
for s, e, n, q in ev[first:first + count]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float('nan')
    print("%8.1f %8.1f  dur %6.1f  gap %5.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, q, n))
    last_end[q] = e
# per-step summary over the whole steady state
fb = [(s, e) for s, e, n, q in ev if 'k_fwd_bwd' in n][20:]
per = [(fb[i + 1][0] - fb[i][0]) / 1e3 for i in range(len(fb) - 1)]
per.sort()
print("step period (k_fwd_bwd start to start): median %.1f us, p10 %.1f, p90 %.1f over %d steps" % (per[len(per) // 2], per[len(per) // 10], per[9 * len(per) // 10], len(per)))
