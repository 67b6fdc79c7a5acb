#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel launch."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
