#!/usr/bin/env python3
"""The frame loop with the NEXT frame's temporal pass riding inside this frame's a-trous launches
(rmd_svgf_frame_atrous_next) against the serial loop, same process, same frames; checks that the outputs agree bit for bit.
    python3 tools/next_probe.py            PROBE_W / PROBE_H (default 3840 x 2160), PROBE_FRAMES"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
from raymarchdenoisercuda_amd.experiments import NextFrameDenoiser  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160)))
FRAMES, WARM, NSEQ = int(os.environ.get("PROBE_FRAMES", 40)), 8, 12
p = rmd.default_params()
p.max_motion_rows = 8
seq = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(NSEQ)]


def run(ahead, frames, outs=None):
    den = NextFrameDenoiser(W, H, params=p)
    out = torch.empty_like(seq[0][0])
    t0 = None
    for f in range(WARM + frames):
        if f == WARM:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        nxt = seq[(f + 1) % NSEQ] if ahead else None
        o = out if outs is None else outs[f]
        den.denoise(*seq[f % NSEQ], out=o, next_frame=nxt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / frames, den


if not rmd.HAS_EXPERIMENTS:
    sys.exit("next_probe needs the experiments build: RMD_LIB_PATH=build/variants/librmd_experiments.so")
n_check = 6
a = [torch.empty_like(seq[0][0]) for _ in range(WARM + n_check)]
b = [torch.empty_like(seq[0][0]) for _ in range(WARM + n_check)]
run(False, n_check, a)
run(True, n_check, b)
same = all(torch.equal(x, y) for x, y in zip(a, b))
del a, b
torch.cuda.empty_cache()
mode = os.environ.get("PROBE_MODE", "both")        # serial | ahead: one form only (for a kernel trace of it)
ms0 = run(False, FRAMES)[0] if mode != "ahead" else float("nan")
ms1 = run(True, FRAMES)[0] if mode != "serial" else float("nan")
ms0b = run(False, FRAMES)[0] if mode == "both" else float("nan")
print(f"next_probe {W}x{H}: serial {ms0:.4f} ms/frame ({W * H / ms0 / 1e3:.0f} Mpix/s), T of the next frame inside the a-trous launches {ms1:.4f} "
      f"({W * H / ms1 / 1e3:.0f} Mpix/s, {(ms0 / ms1 - 1) * 100:+.1f} %), serial again {ms0b:.4f}; outputs bit-identical: {same}")
