#!/usr/bin/env python3
"""Runs only the a-trous iterations on a 4K synthetic frame (steady-state input from 3 frames of
the full pipeline) so that rocprofv3 --pmc passes see the graded kernel alone; prints the average
time per iteration (HIP events on the launch stream).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o name -- python tools/atrous_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160)))
REPS = int(os.environ.get("PROBE_REPS", 5))
VARIANT = int(os.environ.get("PROBE_VARIANT", 0))
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(3)]
for c, nd, m in frames:
    den.denoise(c, nd, m)
torch.cuda.synchronize()
c, nd, m = frames[-1]
desc = den.describe(c, nd, m, den.ping[1])
p.atrous_variant = VARIANT
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(REPS)]
for rep in range(REPS):
    src, dst = den.v_color, den.ping[0]
    ev[rep][0].record()
    for it in range(5):
        rmd.svgf.atrous(desc, p, it, src, dst, 0, H)
        ev[rep][it + 1].record()
        src, dst = dst, (den.ping[1] if dst is den.ping[0] else den.ping[0])
torch.cuda.synchronize()
per_it = [sum(ev[r][i].elapsed_time(ev[r][i + 1]) for r in range(1, REPS)) / max(REPS - 1, 1) * 1e3 for i in range(5)]
print("probe", os.environ.get("RMD_LIB_PATH", "default"), W, H, "us/iteration", [round(t, 1) for t in per_it], "sum", round(sum(per_it), 1))
