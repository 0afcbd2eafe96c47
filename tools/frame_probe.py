#!/usr/bin/env python3
"""Runs whole SVGF frames (T, V, A0..A4) at PROBE_W x PROBE_H so rocprofv3 --kernel-trace --stats
shows the per-kernel split at a size other than bench.py's 4K workload; prints the wall time per
frame measured with HIP events on the denoiser's stream.
    rocprofv3 --kernel-trace --stats -d out -o name -- python3 tools/frame_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080)))
FRAMES = int(os.environ.get("PROBE_FRAMES", 60))
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
inputs = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(8)]
for f in range(8):
    den.denoise(*inputs[f])
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for f in range(FRAMES):
    den.denoise(*inputs[f % 8])
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / FRAMES
print(f"frame_probe {W}x{H}: {ms:.4f} ms/frame, {W * H / ms / 1e3:.1f} Mpix/s over {FRAMES} frames")
