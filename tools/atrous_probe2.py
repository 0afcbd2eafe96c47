#!/usr/bin/env python3
"""Time each a-trous iteration for several variants (HIP events through torch on the current stream).
    python3 tools/atrous_probe2.py [variants...]      default: 3 4"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
variants = [int(v) for v in sys.argv[1:]] or [3, 4]
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(3)]
for c, nd, m in frames:
    den.denoise(c, nd, m)
torch.cuda.synchronize()
c, nd, m = frames[-1]
desc = den.describe(c, nd, m, den.ping[1])
for v in variants:
    p.atrous_variant = v
    src, dst = den.v_color, den.ping[0]
    per = []
    for it in range(5):
        for _ in range(3):
            rmd.svgf.atrous(desc, p, it, src, dst, 0, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            rmd.svgf.atrous(desc, p, it, src, dst, 0, H)
        e1.record()
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) / reps * 1e3)
        src, dst = dst, (den.ping[1] if dst is den.ping[0] else den.ping[0])
    print(f"variant {v}: " + "  ".join(f"{t:6.1f}" for t in per) + f"   sum {sum(per):7.1f} us  ({W}x{H})")
