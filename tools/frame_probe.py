#!/usr/bin/env python3
"""Runs whole SVGF frames (T, V, A0..A4) at PROBE_W x PROBE_H so rocprofv3 --kernel-trace --stats
shows the per-kernel split at a size other than bench.py's 4K workload; prints the wall time per
frame (host clock around a synchronised run of PROBE_FRAMES frames).
    rocprofv3 --kernel-trace --stats -d out -o name -- python3 tools/frame_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080)))
FRAMES = int(os.environ.get("PROBE_FRAMES", 60))
PAN = tuple(float(v) for v in os.environ.get("PROBE_PAN", "1.25,-0.5").split(","))
p = rmd.default_params()
p.max_motion_rows = 8
p.tv_workgroups = int(os.environ.get("PROBE_TV_WG", 0))
p.hist_iteration = int(os.environ.get("PROBE_HIST_IT", 0))
den = rmd.SvgfDenoiser(W, H, params=p, pipelined=os.environ.get("PROBE_PIPELINE", "0") == "1")
inputs = [rmd.svgf.synth_gbuffer(W, H, f, pan=PAN) for f in range(8)]
for f in range(8):
    den.denoise(*inputs[f])
den.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(FRAMES):
    den.denoise(*inputs[f % 8])
den.synchronize()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / FRAMES
flagged = int(den.tile_flags.sum().item())
print(f"frame_probe {W}x{H} pan {PAN}: tiles with short-history pixels {flagged} of {den.tile_flags.numel()}")
print(f"frame_probe {W}x{H}: {ms:.4f} ms/frame, {W * H / ms / 1e3:.1f} Mpix/s over {FRAMES} frames")
