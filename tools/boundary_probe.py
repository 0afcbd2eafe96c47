#!/usr/bin/env python3
"""What a launch boundary of the frame costs.  rocprofv3 reports consecutive dispatches of one queue with Start(k+1) == End(k), so
the boundary (barrier, cache write-back / invalidate, wave launch, the tail behind the last wave) hides INSIDE the reported
durations.  This probe runs the frame's launches back to back up to a-trous iteration i, i = 0..4, with the trace build
(-DRMD_ATROUS_TRACE -DRMD_EXPERIMENTS): the last launch's workgroups record their own start / end times (100 MHz realtime
counter), i.e. the span from the first wave's first instruction to the last wave's last.  Run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- python3 tools/boundary_probe.py
and reduce with  tools/boundary_probe.py --reduce DIR spans.json : per iteration, dispatch duration (rocprofv3) - span (in kernel).
    RMD_NT_OUT=0|1 forces the non-temporal output stores off / on (4K default: on)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) >= 4 and sys.argv[1] == "--reduce":
    import csv
    import glob
    import statistics
    path = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    spans = json.load(open(sys.argv[3]))
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("void ", "").replace("rmd::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    ours = [x for x in rows if x[2].startswith(("svgf_temporal", "atrous_stream"))]
    # runs: T, A0..Ai, each run ends where the next svgf_temporal starts
    runs, cur = [], []
    for s, e, n in ours:
        if n.startswith("svgf_temporal") and cur:
            runs.append(cur)
            cur = []
        cur.append((s, e, n))
    runs.append(cur)
    by_it = {}
    for run in runs:
        it = len(run) - 2
        if it >= 0:
            s, e, n = run[-1]
            gap = (s - run[-2][1]) / 1e3
            by_it.setdefault(it, []).append(((e - s) / 1e3, gap))
    print(f"{path}\\nper a-trous iteration (last launch of a back-to-back run T, A0..Ai; {spans['reps']} runs each, nt_out = {spans['nt_out']}):")
    print("  it  dispatch duration (rocprofv3)   in-kernel span (first wave start -> last wave end)   difference = launch boundary   gap in front")
    tot = 0.0
    for it in sorted(by_it):
        d = [x[0] for x in by_it[it]][-spans["reps"]:]
        g = [x[1] for x in by_it[it]][-spans["reps"]:]
        sp = spans["span_us"][str(it)]
        dm, sm = statistics.median(d), statistics.median(sp)
        tot += dm - sm
        print(f"  {it}   {dm:8.1f} us (min {min(d):.1f})          {sm:8.1f} us (min {min(sp):.1f})                         {dm - sm:6.1f} us            {statistics.median(g):5.2f} us")
    print(f"  sum over the five iterations: {tot:.1f} us per frame inside the reported kernel durations that no wave is running")
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
REPS = int(os.environ.get("PROBE_REPS", 12))
raw = C.CDLL(rmd.LIB_PATH)
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(6)]
out = torch.empty_like(frames[0][0])
for k in range(150):                       # history + clock preconditioning
    den.denoise(*frames[k % 6], out)
torch.cuda.synchronize()
NWG = 8192
buf = np.zeros(NWG * 6, np.uint64)
spans = {str(i): [] for i in range(5)}
for rep in range(REPS + 2):
    for it in range(5):
        c, nd, m = frames[(rep + it) % 6]
        desc = den.describe(c, nd, m, out)
        rmd.check(rmd.lib.rmd_svgf_frame_tv(C.byref(desc), C.byref(p), 0, H, None))
        src, pp = den.v_color, 0
        for k in range(it + 1):            # A0 .. A_it back to back behind T+V, with rmd_svgf_frame's plane routing
            if k == 4:
                dst = out
            elif k == p.hist_iteration:
                dst = den.hist_color[den.cur ^ 1]
            else:
                dst, pp = den.ping[pp], pp ^ 1
            rmd.svgf.atrous(desc, p, k, src, dst, 0, H)
            src = dst
        assert raw.rmd_debug_atrous_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0       # synchronises; the LAST launch's records
        rec = buf.reshape(NWG, 6)
        t = rec[(rec[:, 0] > 0) & ((rec[:, 5] >> 32) == (1 << it))]       # the last launch's own records (a record is keyed by workgroup id:
        if rep >= 2:                                                         # an earlier launch with more workgroups leaves some behind)
            spans[str(it)].append(float(t[:, 1].max() - t[:, 0].min()) / 100.0)
json.dump({"reps": REPS, "nt_out": os.environ.get("RMD_NT_OUT", "default (on at 4K)"), "span_us": spans},
          open(os.environ.get("PROBE_OUT", "spans.json"), "w"))
print({k: round(float(np.median(v)), 1) for k, v in spans.items()})
