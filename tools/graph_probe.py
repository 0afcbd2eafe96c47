#!/usr/bin/env python3
"""Does replaying the frame's six launches from a hipGraph change the frame time?  Two frames (the history planes
ping-pong) are captured on a side stream with torch.cuda.CUDAGraph and replayed; the same two frames are timed eagerly
before and after, in the same process.
    python3 tools/graph_probe.py            PROBE_W / PROBE_H (default 1920 x 1080), PROBE_FRAMES"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080)))
PAIRS = int(os.environ.get("PROBE_FRAMES", 60)) // 2
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
a, b = rmd.svgf.synth_gbuffer(W, H, 0), rmd.svgf.synth_gbuffer(W, H, 1)
out = torch.empty_like(a[0])


def pair():
    den.denoise(*a, out=out)
    den.denoise(*b, out=out)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (2 * n)


for _ in range(4):
    pair()
eager0 = timed(pair, PAIRS)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    pair()                                     # first use of the side stream
    side.synchronize()
    g.capture_begin()
    pair()
    g.capture_end()
torch.cuda.current_stream().wait_stream(side)
ref = out.clone()
for _ in range(4):
    g.replay()
graph = timed(g.replay, PAIRS)
same = torch.equal(out, ref)
eager1 = timed(pair, PAIRS)
print(f"graph_probe {W}x{H}: eager {eager0:.4f} ms/frame, graph replay {graph:.4f} ms/frame ({(eager0 / graph - 1) * 100:+.1f} %), eager again {eager1:.4f}; "
      f"replayed output equals the eager one: {same}")
