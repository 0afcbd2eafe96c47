#!/usr/bin/env python3
"""Debug aid: where does a-trous variant 4 (pair stream kernel) differ from variant 5 (its direct form)?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

for (W, H) in [(64, 48), (300, 70), (523, 301), (1920, 1080)]:
    c, nd, m = rmd.svgf.synth_gbuffer(W, H, 3)
    c[..., 3] = torch.rand((H, W), device="cuda") * 0.3
    d = rmd.svgf.frame_desc(W, H, nd=nd)
    p = rmd.default_params()
    src = c
    for it in range(5):
        outs = {}
        for v in (1, 5, 4):
            p.atrous_variant = v
            o = torch.full_like(c, float("nan"))
            rmd.svgf.atrous(d, p, it, src, o, 0, H)
            outs[v] = o
        torch.cuda.synchronize()
        a, b, r = outs[4].cpu().numpy(), outs[5].cpu().numpy(), outs[1].cpu().numpy()
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        e51 = np.abs(b - r) / (1 + np.abs(r))
        print(f"{W}x{H} it {it}: v4!=v5 at {len(bad)} values; v5 vs v1 max scaled err {np.nanmax(e51):.2e}; nan in v4 {np.isnan(a).sum()} v5 {np.isnan(b).sum()}")
        if len(bad):
            ys, xs, cs = bad[:, 0], bad[:, 1], bad[:, 2]
            print("   rows", np.unique(ys)[:20], "... cols", np.unique(xs)[:40], "chan", np.unique(cs))
            for y, x, ch in bad[:6]:
                print(f"   ({y},{x},{ch}): v4 {a[y, x, ch]:.7g} v5 {b[y, x, ch]:.7g} v1 {r[y, x, ch]:.7g}")
        src = outs[5]
