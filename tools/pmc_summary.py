#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv output directory: mean counter value per kernel name."""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
