#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace run as a timeline: python scripts/ktimeline.py <dir> <anchor kernel substring> [k-th occurrence]"""
import csv, glob, sys
d, anchor = sys.argv[1], sys.argv[2]
kth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = idx[kth], idx[kth + 1]
t0 = int(rows[i0]["Start_Timestamp"])
print("step: %.1f us between two launches of %s" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, anchor))
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f -> %8.1f  (%5.1f us)  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
