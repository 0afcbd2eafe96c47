#!/usr/bin/env python3
"""What one rank of an N-GPU row-strip run costs WITHOUT its halo exchange: the strip of rank R of a
7680 x (1080 N) frame (bench.py's weak-scaling workload) on this one GPU, next to the 3840x2160 frame
of the N = 1 run (same pixel count).  The ratio is the weak-scaling efficiency the redundant halo rows
alone allow; the exchange itself (2.6 MB per neighbour and frame) is not in it.
    python3 tools/strip_probe.py [N [R]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
from raymarchdenoisercuda_amd import sharding  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else N // 2
FRAMES, WARM = 40, 8
p = rmd.default_params()
p.max_motion_rows = 8


SYNC = None


def run(make, px):
    global SYNC
    SYNC = None
    den, frames = make()
    sync = SYNC if SYNC else (lambda: None)
    for f in range(WARM):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(WARM, WARM + FRAMES):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / FRAMES
    return ms, px / ms / 1e3


def single():
    d = rmd.SvgfDenoiser(3840, 2160, params=p)
    return d.denoise, [rmd.svgf.synth_gbuffer(3840, 2160, f) for f in range(WARM + FRAMES)]


def strip():
    sd = sharding.ShardedDenoiser(7680, 1080 * N, params=p, rank=R, world=N, pipelined=os.environ.get("PROBE_PIPELINE", "0") == "1")
    global SYNC
    SYNC = sd.synchronize
    print(f"rank {R} of {N}: output rows [{sd.plan.row0}, {sd.plan.row1}), buffer rows [{sd.plan.buf_row0}, {sd.plan.buf_row0 + sd.plan.buf_rows})")
    return sd.denoise, [sd.synth(f) for f in range(WARM + FRAMES)]


ms1, mp1 = run(single, 3840 * 2160)
msn, mpn = run(strip, 7680 * 1080)
print(f"3840x2160 frame      : {ms1:.4f} ms/frame  {mp1:.0f} Mpix/s")
print(f"7680x1080 strip+halos: {msn:.4f} ms/frame  {mpn:.0f} Mpix/s  -> efficiency bound {ms1 / msn:.3f} (x{N} GPUs: {N * ms1 / msn:.2f}x)")
