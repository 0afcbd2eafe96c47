#!/usr/bin/env python3
"""tools/pmc_traffic.py <pmc dir> <out.json> [shader GHz] [pmc_frame dir] -- HBM bytes per launch and VALU issue figures of the
a-trous kernels from the rocprofv3 --pmc passes of tools/pmc_passes.sh (one counter set per pass).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are in
KiB; FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.  SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles.

ONE denominator for every "busy" figure (round 2 had two: 0.72 in this file, 0.90 in DESIGN.md): the cycles of a launch are
its DURATION (start / end timestamps of the dispatch in the same counter pass) x the shader clock measured INSIDE the kernel
(s_memtime / s_memrealtime of the trace build, tools/atrous_trace.py: 2.02-2.07 GHz under this load; third argument,
default 2.05).  GRBM_GUI_ACTIVE / 8 -- what round 2's 0.72 divided by -- reads 2.3-2.5 GHz on these 0.13 ms dispatches
(the guide: "the quotient reads high on dispatches shorter than about 0.3 ms") and is kept only as `grbm_cycles`."""
import csv, glob, json, os, sys, collections

d, out = sys.argv[1], sys.argv[2]
ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.05
WANT = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "atrous_stream_kernel" in row["Kernel_Name"] and row["Counter_Name"] in WANT:
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "SQ_ACTIVE_INST_VALU":          # durations of the launches of THIS pass
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
mean = lambda v: sum(v) / len(v)  # noqa: E731
per, valu = {}, {}
for k, v in sorted(acc.items()):
    fetch, write = mean(v["FETCH_SIZE"]), mean(v["WRITE_SIZE"])
    per[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes": int((2.0 * fetch + write) * 1024)}
    if all(c in v for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")) and dur[k]:
        us = mean(dur[k])
        cyc = us * ghz * 1e3                                           # shader cycles of the launch
        valu[k] = {"valu_wave_insts": int(mean(v["SQ_INSTS_VALU"])), "launch_us_in_this_pass": round(us, 1), "kernel_cycles": int(cyc),
                   "valu_busy_frac": round(mean(v["SQ_ACTIVE_INST_VALU"]) * 4 / (cyc * 1024), 3),
                   "cycles_per_valu_inst": round(mean(v["SQ_ACTIVE_INST_VALU"]) * 4 / mean(v["SQ_INSTS_VALU"]), 2),
                   "resident_waves_per_simd": round(mean(v["SQ_WAVE_CYCLES"]) * 4 / cyc / 1024, 2)}
        if "SQ_WAIT_ANY" in v:
            valu[k]["wave_time_parked_frac"] = round(mean(v["SQ_WAIT_ANY"]) / mean(v["SQ_WAVE_CYCLES"]), 3)
        if "GRBM_GUI_ACTIVE" in v:
            valu[k]["grbm_cycles"] = int(mean(v["GRBM_GUI_ACTIVE"]) / 8.0)
            valu[k]["grbm_implied_ghz"] = round(mean(v["GRBM_GUI_ACTIVE"]) / 8.0 / us / 1e3, 2)
avg = int(sum(p["hbm_bytes"] for p in per.values()) / max(1, len(per)))
doc = {"atrous_hbm_bytes_per_launch": avg, "workload": "3840x2160, tools/atrous_probe.py", "per_kernel": per,
       "method": "rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
if valu:
    doc["valu_issue"] = {
        "per_kernel": valu, "shader_ghz": ghz,
        "avg_valu_busy_frac": round(mean([x["valu_busy_frac"] for x in valu.values()]), 3),
        "avg_resident_waves_per_simd": round(mean([x["resident_waves_per_simd"] for x in valu.values()]), 2),
        "method": "SQ_ACTIVE_INST_VALU (quad-cycles) x4 / (launch duration in the same pass x the in-kernel shader clock x 1024 SIMDs); "
                  "SQ_WAVE_CYCLES (quad-cycles) x4 / the same cycles / 1024 SIMDs; 3 waves per SIMD is the maximum for this kernel"}
# T+V (HBM-bound): FETCH_SIZE / WRITE_SIZE of the frame loop (tools/pmc_frame.sh), MEDIANS over the launches (the frames without
# history run V everywhere); optional 4th argument = that directory
if len(sys.argv) > 4:
    tv = collections.defaultdict(list)
    tvdur = []
    for f in sorted(glob.glob(os.path.join(sys.argv[4], "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            if "svgf_temporal_variance_kernel" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tv[row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "WRITE_SIZE":
                    tvdur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    if tv["FETCH_SIZE"] and tv["WRITE_SIZE"]:
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        doc["temporal_variance"] = {
            "FETCH_SIZE_KiB_median": med(tv["FETCH_SIZE"]), "WRITE_SIZE_KiB_median": med(tv["WRITE_SIZE"]),
            "hbm_bytes_per_launch": int((2.0 * med(tv["FETCH_SIZE"]) + med(tv["WRITE_SIZE"])) * 1024),
            "launch_us_in_the_write_pass_median": round(med(tvdur), 1),
            "algorithmic_bytes_per_launch": 106 * 3840 * 2160,
            "note": "the x2 of FETCH_SIZE is calibrated for 16 B/lane reads (64 of this kernel's 81 read bytes per pixel); medians of the frame loop"}
json.dump(doc, open(out, "w"), indent=1)
print("atrous_hbm_bytes_per_launch", avg, "algorithmic", 48 * 3840 * 2160)
if valu:
    for k, x in valu.items():
        print(k, x)
