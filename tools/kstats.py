#!/usr/bin/env python3
"""Print a rocprofv3 kernel-stats CSV compactly: name, calls, average ms, share.  usage: tools/kstats.py file.csv [runs]"""
import csv, sys
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in csv.DictReader(open(sys.argv[1])):
    ms = float(r['AverageNs']) / 1e6
    if float(r['Percentage']) < 0.3: continue
    print("  %-58s calls/run %6.1f avg %9.3f ms  per-run %9.2f ms  %5.1f %%" % (r['Name'].split('(')[0][-58:], int(r['Calls']) / runs, ms, ms * int(r['Calls']) / runs, float(r['Percentage'])))
