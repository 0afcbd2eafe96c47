#!/usr/bin/env python3
"""Prints calls / average / min / max (us) per kernel from a rocprofv3 `*_kernel_stats.csv`.
    python3 tools/kstats.py <dir or csv> [name filter ...]
"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[-1]
filters = sys.argv[2:]
for r in csv.DictReader(open(path)):
    name = r["Name"].replace("void ", "").replace("rmd::", "")
    if filters and not any(f in name for f in filters):
        continue
    if not filters and float(r["Percentage"]) < 0.5:
        continue
    print(f"{name[:58]:58s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e3:8.1f}  min {int(r['MinNs']) / 1e3:8.1f}  max {int(r['MaxNs']) / 1e3:8.1f}  {float(r['Percentage']):5.1f}%")
