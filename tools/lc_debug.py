#!/usr/bin/env python3
"""Debug aid: a-trous variant 7 (loader/consumer pair kernel) against variant 5 (its direct form), bit for bit."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

V = int(os.environ.get("LC_VARIANT", 7))
total_bad = 0
for (W, H) in [(64, 48), (300, 70), (523, 301), (1920, 1080), (3840, 2160)]:
    c, nd, m = rmd.svgf.synth_gbuffer(W, H, 3)
    c[..., 3] = torch.rand((H, W), device="cuda") * 0.3
    if W > 400:
        nd[H // 3:H // 3 + 9, W // 4:W // 4 + 37, :3] = 0        # background pixels: the zero-aware tap path
    d = rmd.svgf.frame_desc(W, H, nd=nd)
    p = rmd.default_params()
    src = c
    for it in range(5):
        outs = {}
        for v in (5, V):
            p.atrous_variant = v
            o = torch.full_like(c, float("nan"))
            rmd.svgf.atrous(d, p, it, src, o, 0, H)
            outs[v] = o
        torch.cuda.synchronize()
        a, b = outs[V].cpu().numpy(), outs[5].cpu().numpy()
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        total_bad += len(bad)
        print(f"{W}x{H} it {it}: v{V}!=v5 at {len(bad)} values; nan in v{V} {np.isnan(a).sum()} v5 {np.isnan(b).sum()}", flush=True)
        if len(bad):
            ys, xs, cs = bad[:, 0], bad[:, 1], bad[:, 2]
            print("   rows", np.unique(ys)[:20], "... cols", np.unique(xs)[:40], "chan", np.unique(cs))
            for y, x, ch in bad[:6]:
                print(f"   ({y},{x},{ch}): v{V} {a[y, x, ch]:.7g} v5 {b[y, x, ch]:.7g}")
        src = outs[5]
print("TOTAL mismatches", total_bad)
sys.exit(1 if total_bad else 0)
