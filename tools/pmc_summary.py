#!/usr/bin/env python3
"""Reduces rocprofv3 --pmc CSVs (tools/pmc_passes.sh) to per-kernel averages.
usage: tools/pmc_summary.py <dir> [--all] [--median] [--json out.json]"""
import csv, glob, json, os, sys, collections

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*_kernel_trace.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
out = {}
for k in sorted(acc):
    if "atrous" not in k and "--all" not in sys.argv:
        continue
    vals = {c: sum(v) / len(v) for c, v in acc[k].items()}
    vals["dur_us(profiled)"] = sum(dur[k]) / max(1, len(dur[k]))
    if "--median" in sys.argv:          # frame loops: the warm-up frames (no history: V everywhere) are in the means, not in the medians
        for c, v in acc[k].items():
            vals[c + " (median)"] = sorted(v)[len(v) // 2]
        if dur[k]:
            vals["dur_us(profiled) (median)"] = sorted(dur[k])[len(dur[k]) // 2]
    out[k] = vals
    print(k)
    for c, v in sorted(vals.items()):
        print(f"    {c:28s} {v:18.1f}")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
