#!/usr/bin/env python3
"""What one rank of an N-GPU row-strip run of the 8K frame costs WITHOUT its halo exchange: the strip
of rank R (4320/N rows + the redundant rows later passes tap) of the 7680x4320 frame of bench.py's
N > 1 workload (BASELINE configs[3]) on this one GPU, next to the unsharded 8K frame.  T(frame) /
T(strip) is the speed-up the redundant rows alone allow (strong scaling, upper bound); the exchange
itself (2.6 MB per neighbour and frame) is not in it.
    python3 tools/strip_probe.py [N [R]]        R defaults to an interior rank (N // 2)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
from raymarchdenoisercuda_amd import sharding  # noqa: E402

W, H = 7680, 4320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RANKS = [int(sys.argv[2])] if len(sys.argv) > 2 else sorted({0, N // 2})
FRAMES, WARM = 24, 6
p = rmd.default_params()
p.max_motion_rows = 8


def run(den, frames, sync):
    for f in range(WARM):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(WARM, WARM + FRAMES):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / FRAMES


d = rmd.SvgfDenoiser(W, H, params=p)
seq = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(8)]
ms1 = run(d.denoise, seq, lambda: None)
del d, seq
torch.cuda.empty_cache()
print(f"7680x4320 frame, one GPU      : {ms1:.4f} ms/frame  {W * H / ms1 / 1e3:.0f} Mpix/s")
for R in RANKS:
    sd = sharding.ShardedDenoiser(W, H, params=p, rank=R, world=N, pipelined=os.environ.get("PROBE_PIPELINE", "0") == "1")
    seq = [sd.synth(f) for f in range(WARM + FRAMES)]
    msn = run(sd.denoise, seq, sd.synchronize)
    pl = sd.plan
    print(f"rank {R} of {N}: rows [{pl.row0}, {pl.row1}) + buffer [{pl.buf_row0}, {pl.buf_row0 + pl.buf_rows}): "
          f"{msn:.4f} ms/frame -> speed-up bound {ms1 / msn:.2f}x of {N} (efficiency {ms1 / msn / N:.3f})")
    del sd, seq
    torch.cuda.empty_cache()
