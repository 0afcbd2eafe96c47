#!/usr/bin/env python3
"""tools/pmc_traffic.py <pmc dir> <out.json> — HBM bytes per launch and VALU issue figures of the
a-trous kernels from the rocprofv3 --pmc passes of tools/pmc_passes.sh (one counter set per pass).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE
and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv, glob, json, os, sys, collections

d, out = sys.argv[1], sys.argv[2]
WANT = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "atrous_stream_kernel" in row["Kernel_Name"] and row["Counter_Name"] in WANT:
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = lambda v: sum(v) / len(v)  # noqa: E731
per, valu = {}, {}
for k, v in sorted(acc.items()):
    fetch, write = mean(v["FETCH_SIZE"]), mean(v["WRITE_SIZE"])
    per[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes": int((2.0 * fetch + write) * 1024)}
    if all(c in v for c in WANT[2:]):
        cyc = mean(v["GRBM_GUI_ACTIVE"]) / 8.0
        valu[k] = {"valu_wave_insts": int(mean(v["SQ_INSTS_VALU"])),
                   "valu_busy_frac": round(mean(v["SQ_ACTIVE_INST_VALU"]) * 4 / (cyc * 1024), 3),
                   "cycles_per_valu_inst": round(mean(v["SQ_ACTIVE_INST_VALU"]) * 4 / mean(v["SQ_INSTS_VALU"]), 2),
                   "resident_waves_per_simd": round(mean(v["SQ_WAVE_CYCLES"]) * 4 / cyc / 1024, 2), "kernel_cycles": int(cyc)}
avg = int(sum(p["hbm_bytes"] for p in per.values()) / max(1, len(per)))
doc = {"atrous_hbm_bytes_per_launch": avg, "workload": "3840x2160, tools/atrous_probe.py", "per_kernel": per,
       "method": "rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
if valu:
    doc["valu_issue"] = {
        "per_kernel": valu,
        "avg_valu_busy_frac": round(mean([x["valu_busy_frac"] for x in valu.values()]), 3),
        "avg_resident_waves_per_simd": round(mean([x["resident_waves_per_simd"] for x in valu.values()]), 2),
        "method": "SQ_ACTIVE_INST_VALU (quad-cycles) x4 / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs); SQ_WAVE_CYCLES (quad-cycles) x4 / "
                  "kernel cycles / 1024 SIMDs; 3 waves per SIMD is the maximum for this kernel"}
json.dump(doc, open(out, "w"), indent=1)
print("atrous_hbm_bytes_per_launch", avg, "algorithmic", 48 * 3840 * 2160)
