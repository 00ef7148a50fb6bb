#!/usr/bin/env python3
"""Per-kernel per-launch averages of a rocprofv3 counter_collection.csv.  usage: pmc_summary.py FILE.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
