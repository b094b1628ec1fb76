#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter).  usage: tools/pmc_quick.py DIR [kernel-substring]"""
import csv, glob, sys, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        d = (f, r["Dispatch_Id"])
        if d not in seen: seen.add(d); calls[(k, f)] += 1
for (k, c), v in sorted(tot.items()):
    print("%-42s %-28s %.4g" % (k, c, v))
