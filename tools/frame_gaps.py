#!/usr/bin/env python3
"""Where a frame's time goes BETWEEN its kernels: from a rocprofv3 --kernel-trace CSV (start / end timestamps per
dispatch) of the bench loop, per steady-state frame: sum of kernel durations, sum of the gaps between consecutive
kernels of the frame, and the gap in front of each kernel (median over frames).
    python3 tools/frame_gaps.py <dir or *_kernel_trace.csv>"""
import csv
import glob
import os
import statistics
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("void ", "").replace("rmd::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0]))
rows.sort()
# a frame = temporal kernel .. the a-trous launch with step 16
frames, cur = [], None
for s, e, n in rows:
    if n.startswith("svgf_temporal"):                      # svgf_temporal_kernel, or the fused svgf_temporal_variance_kernel
        cur = [(s, e, n)]
    elif cur is not None and (n.startswith("svgf_variance") or n.startswith("atrous_stream_kernel")):
        cur.append((s, e, n))
        if n.startswith("atrous_stream_kernel<16"):
            if len(cur) in (6, 7):
                frames.append(cur)
            cur = None
    else:
        cur = None
frames = frames[len(frames) // 4:]                      # skip warm-up
if not frames:
    sys.exit("no complete frames found")
busy = [sum(e - s for s, e, _ in f) / 1e3 for f in frames]
span = [(f[-1][1] - f[0][0]) / 1e3 for f in frames]
period = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(frames, frames[1:]) if b[0][0] - a[0][0] < 3 * (a[-1][1] - a[0][0])]
print(f"{len(frames)} frames: kernel time {statistics.median(busy):.1f} us, first start -> last end {statistics.median(span):.1f} us, "
      f"frame period {statistics.median(period) if period else float('nan'):.1f} us (medians)")
for k in range(len(frames[0])):
    dur = statistics.median((f[k][1] - f[k][0]) / 1e3 for f in frames)
    gap = statistics.median(((f[k][0] - f[k - 1][1]) / 1e3) for f in frames) if k else float("nan")
    print(f"  {frames[0][k][2][:40]:40s} duration {dur:7.1f} us   gap in front {gap:6.1f} us")
if period:
    between = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(frames, frames[1:]) if b[0][0] - a[0][0] < 3 * (a[-1][1] - a[0][0])]
    print(f"  gap between frames (A4 end -> next T start) {statistics.median(between):.1f} us")
