#!/usr/bin/env python3
"""tools/pmc_traffic.py <pmc dir> <out.json> — HBM bytes per launch of the a-trous kernels from the
rocprofv3 --pmc passes of tools/pmc_passes.sh (FETCH_SIZE and WRITE_SIZE come from separate passes).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950: both
counters are in KiB; FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import csv, glob, json, os, sys, collections

d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "atrous_stream_kernel" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
per = {}
for k, v in sorted(acc.items()):
    fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
    write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    per[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes": int((2.0 * fetch + write) * 1024)}
avg = int(sum(p["hbm_bytes"] for p in per.values()) / max(1, len(per)))
json.dump({"atrous_hbm_bytes_per_launch": avg, "workload": "3840x2160, tools/atrous_probe.py", "per_kernel": per,
           "method": "rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"},
          open(out, "w"), indent=1)
print("atrous_hbm_bytes_per_launch", avg, "algorithmic", 48 * 3840 * 2160)
