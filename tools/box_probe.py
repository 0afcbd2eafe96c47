#!/usr/bin/env python3
"""Throughput of the reference-API kernels (uchar4 box mean and the weighted modes) at 4K."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160)))
g = torch.Generator(device="cuda").manual_seed(1)
planes = {k: torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda", generator=g) for k in ("render", "normal", "albedo")}
out = torch.empty_like(planes["render"])
bufs = (torch.empty_like(out), torch.empty_like(out))
frame = rmd.make_gbuffer(planes["render"], out, bufs[0], bufs[1], normal=planes["normal"], albedo=planes["albedo"])
cases = [("baseline AVERAGE r=2 depth=1", rmd.filterKernelBaseline, rmd.FilterParams(radius=2, depth=1)),
         ("tiled    AVERAGE r=2 depth=1", rmd.filterKernelTiled, rmd.FilterParams(radius=2, depth=1)),
         ("tiled    AVERAGE r=2 depth=5", rmd.filterKernelTiled, rmd.FilterParams(radius=2, depth=5)),
         ("tiled    AVERAGE r=5 depth=1", rmd.filterKernelTiled, rmd.FilterParams(radius=5, depth=1)),
         ("tiled    AVERAGE r=8 depth=1", rmd.filterKernelTiled, rmd.FilterParams(radius=8, depth=1)),
         ("tiled    AVERAGE r=16 depth=1", rmd.filterKernelTiled, rmd.FilterParams(radius=16, depth=1)),
         ("tiled    AVERAGE r=32 depth=1", rmd.filterKernelTiled, rmd.FilterParams(radius=32, depth=1)),
         ("tiled    GAUSSIAN r=2", rmd.filterKernelTiled, rmd.FilterParams(type=rmd.FilterParams.GAUSSIAN, radius=2, sigmaSpace=1.5)),
         ("tiled    CROSS r=2", rmd.filterKernelTiled, rmd.FilterParams(type=rmd.FilterParams.CROSS, radius=2, sigmaSpace=1.5, sigmaColor=0.2, sigmaNormal=0.2, sigmaAlbedo=0.2)),
         ("tiled    WAVELET depth=5", rmd.filterKernelTiled, rmd.FilterParams(type=rmd.FilterParams.WAVELET, depth=5, sigmaColor=0.2, sigmaNormal=0.2, sigmaAlbedo=0.2))]
for name, fn, p in cases:
    for _ in range(3):
        fn(frame, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(frame, p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:32s} {ms * 1e3:8.1f} us  {W * H / ms / 1e3:9.0f} Mpix/s  {8.0 * max(p.depth, 1) * W * H / ms / 1e6:7.0f} GB/s algorithmic")
