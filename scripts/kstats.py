#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats output dir: python scripts/kstats.py gpurun_out/p4"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 9]:
    print(r['Name'][:52].ljust(52), r['Calls'].rjust(5), "%7.1f us" % (float(r['AverageNs']) / 1e3), r['Percentage'])
