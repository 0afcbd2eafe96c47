#!/usr/bin/env python3
"""The 4K animated Cornell sequence (BASELINE configs[4]) through ONE of the three forms bench.py compares, for a rocprofv3
kernel trace of that form alone:
    PROBE_FORM=fused    one rmd_svgf_gbuffer_frame call per frame (uchar4 in -> uchar4 out, 6 launches)      [default]
    PROBE_FORM=chain    the eight-call chain on the same bytes (11 launches)
    PROBE_FORM=float    float planes in / out (6 launches; conversions outside the loop)
    PROBE_W / PROBE_H (3840 x 2160), PROBE_FRAMES (40)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
import bench  # noqa: E402

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
FRAMES, WARM = int(os.environ.get("PROBE_FRAMES", 40)), 6
FORM = os.environ.get("PROBE_FORM", "fused")
EPS = 1.0 / 255.0
pan = (2.25, 1.5)
p = rmd.default_params()
p.max_motion_rows = 8
seq = bench.cornell_u8_sequence(torch, W, H, WARM + FRAMES, pan)
motion = torch.empty((H, W, 2), dtype=torch.float32, device="cuda")
motion[..., 0], motion[..., 1] = -pan[0], -pan[1]
npx = W * H
if FORM == "fused":
    den = rmd.GBufferDenoiser(W, H, params=p, albedo_eps=EPS)
    out = torch.empty_like(seq[0][0])
    step = lambda f: den.frame(seq[f][0], seq[f][1], seq[f][2], out, motion)          # noqa: E731
else:
    den = rmd.SvgfDenoiser(W, H, params=p)
    f4 = lambda: torch.empty((H, W, 4), dtype=torch.float32, device="cuda")            # noqa: E731
    color, alb, out_f32, nds, out = f4(), f4(), f4(), [f4(), f4()], torch.empty_like(seq[0][0])
    if FORM == "chain":
        def step(f):
            rn, al, nm = seq[f]
            nd = nds[f & 1]
            rmd.check(rmd.lib.rmd_convert_u8_to_f32(rn.data_ptr(), color.data_ptr(), npx, 0, 0.0, None))
            rmd.check(rmd.lib.rmd_convert_u8_to_f32(al.data_ptr(), alb.data_ptr(), npx, 0, 0.0, None))
            rmd.check(rmd.lib.rmd_convert_u8_to_f32(nm.data_ptr(), nd.data_ptr(), npx, 1, -1.0, None))
            rmd.check(rmd.lib.rmd_demodulate(color.data_ptr(), alb.data_ptr(), color.data_ptr(), npx, EPS, None))
            den.denoise(color, nd, motion, out_f32)
            rmd.check(rmd.lib.rmd_convert_f32_to_u8(out_f32.data_ptr(), alb.data_ptr(), out.data_ptr(), npx, None))
    else:
        fseq = []
        for rn, al, nm in seq:
            c = rmd.svgf.convert_u8_to_f32(rn, False, 0.0)
            a = rmd.svgf.convert_u8_to_f32(al, False, 0.0)
            rmd.svgf.demodulate(c, a, EPS, out=c)
            fseq.append((c, rmd.svgf.convert_u8_to_f32(nm, True, -1.0)))
        step = lambda f: den.denoise(fseq[f][0], fseq[f][1], motion, out_f32)          # noqa: E731
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.08:           # clock preconditioning (bench.py)
    for f in range(WARM + FRAMES):
        step(f)
    torch.cuda.synchronize()
den.reset_history()
for f in range(WARM):
    step(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(WARM, WARM + FRAMES):
    step(f)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{FORM}: {FRAMES} frames of {W}x{H}: {dt / FRAMES * 1e3:.4f} ms per frame, {FRAMES / dt:.1f} fps, {W * H * FRAMES / dt / 1e6:.0f} Mpix/s")
