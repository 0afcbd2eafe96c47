#!/usr/bin/env python3
"""Where a frame's time goes, from a rocprofv3 --kernel-trace CSV (start / end timestamp per dispatch) of a frame loop.
Per frame (a frame = the temporal launch .. the last a-trous launch in front of the next temporal launch):
  period (this T start -> next T start), kernel time (sum of the dispatch durations), and period - kernel time = the SUM of
  the gaps (between consecutive launches of the frame + from its last launch to the next frame), the mean gap, and the three
  largest gaps with their position.
Then, over all frames of the run in order, the per-launch durations: on a cold GPU they DRIFT (the a-trous launches are
clock-bound and the shader clock settles over the first ~50 ms of load), so a median over one subset of frames must not be
compared with a median over another -- which is all that round 3's "40-55 us per frame in no kernel" was.
    python3 tools/frame_gaps.py <dir or *_kernel_trace.csv> [--ramp] [--frames=LO:HI]     (LO:HI = the frames of the trace to analyse,
    e.g. the timed region of bench.py without its roofline loop, whose HIP-event timers sit between the launches)
Note on rocprofv3's timestamps: consecutive dispatches of one queue are reported with Start(k+1) == End(k) to the nanosecond
whenever k+1 was already queued, so whatever the launch boundary costs (barrier, cache write-back / invalidate, wave launch) is
INSIDE the reported duration of k+1; only a host-side stall or a stream operation (an event record: ~6 us) shows up as a gap."""
import csv
import glob
import os
import statistics
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path = args[0]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("void ", "").replace("rmd::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0]))
rows.sort()
ours = ("svgf_temporal", "svgf_variance", "atrous_", "u8_to_f32", "f32_to_u8", "demodulate")
frames, cur = [], None
for s, e, n in rows:
    if n.startswith("svgf_temporal"):
        if cur:
            frames.append(cur)
        cur = [(s, e, n)]
    elif cur is not None and n.startswith(ours):
        cur.append((s, e, n))
    elif cur is not None:
        frames.append(cur)
        cur = None
if cur:
    frames.append(cur)
shape = statistics.mode(len(f) for f in frames)
frames = [f for f in frames if len(f) == shape]
if len(frames) < 4:
    sys.exit("no frame loop found")

print(f"{path}\n{len(frames)} frames of {shape} launches")
table = []
for a, b in zip(frames, frames[1:]):
    period = (b[0][0] - a[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in a) / 1e3
    if period > 3 * busy:                # not consecutive frames of one loop
        continue
    gaps = [((a[k][0] - a[k - 1][1]) / 1e3, f"in front of launch {k} ({a[k][2][:28]})") for k in range(1, shape)]
    gaps.append(((b[0][0] - a[-1][1]) / 1e3, "last launch -> next frame's T"))
    table.append((period, busy, gaps))
sel = [a for a in sys.argv[1:] if a.startswith("--frames=")]
if sel:
    lo, hi = (int(v) if v else None for v in sel[0].split("=")[1].split(":"))
    steady = table[lo:hi]
else:
    steady = table[len(table) // 4:]
print(f"analysed frames ({sel[0] if sel else 'last three quarters'}, {len(steady)}): period {statistics.mean(p for p, _, _ in steady):.1f} us mean / "
      f"{statistics.median(p for p, _, _ in steady):.1f} median; kernel time {statistics.mean(b for _, b, _ in steady):.1f} mean / "
      f"{statistics.median(b for _, b, _ in steady):.1f} median")
d = [p - b for p, b, _ in steady]
print(f"period - kernel time PER FRAME: mean {statistics.mean(d):.2f} us, median {statistics.median(d):.2f}, min {min(d):.2f}, max {max(d):.2f}  "
      f"(= sum of the {shape} gaps of the frame; mean gap {statistics.mean(d) / shape:.2f} us)")
pos = {}
for _, _, gaps in steady:
    for g, where in sorted(gaps, reverse=True)[:3]:
        if g > 0.05:
            pos.setdefault(where, []).append(g)
if not pos:
    print("no gap above 0.05 us anywhere: every launch was queued behind its predecessor")
for where, g in sorted(pos.items(), key=lambda kv: -sum(kv[1])):
    print(f"  among the three largest gaps of a frame in {len(g):3d} of {len(steady)} frames: {where}: mean {statistics.mean(g):.2f} us, max {max(g):.2f}")
print("per-launch durations, median over the steady frames:")
for k in range(shape):
    fs = frames[lo:hi] if sel else frames[len(frames) // 4:]
    print(f"  {frames[0][k][2][:44]:44s} {statistics.median((f[k][1] - f[k][0]) / 1e3 for f in fs):7.1f} us")
if "--ramp" in sys.argv:
    print("frame  " + "  ".join(f"L{k}" .rjust(6) for k in range(shape)) + "     sum   (durations in us, every frame of the trace in order)")
    for i, f in enumerate(frames):
        print(f"{i:5d}  " + "  ".join(f"{(e - s) / 1e3:6.1f}" for s, e, _ in f) + f"  {sum(e - s for s, e, _ in f) / 1e3:7.1f}")
