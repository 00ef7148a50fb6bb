#!/usr/bin/env python3
"""Rows of a rocprofv3 `*_kernel_stats.csv` whose kernel name contains a substring: name, calls, average ns.  usage: kernel_stats_grep.py FILE SUBSTRING"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}')
