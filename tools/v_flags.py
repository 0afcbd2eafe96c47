#!/usr/bin/env python3
"""How many 64x4 tiles T flags for V in the steady state of bench.py's synthetic sequence, and how they are laid out."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = 3840, 2160
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p, debug=True)
for f in range(10):
    c, nd, m = rmd.svgf.synth_gbuffer(W, H, f)
    den.denoise(c, nd, m)
    torch.cuda.synchronize()
    tx, ty = (W + 63) // 64, (H + 3) // 4
    fl = den.tile_flags[:tx * ty].reshape(ty, tx).cpu().numpy() != 0
    hist = den.hist_len[den.cur]
    short = (hist < p.var_h_threshold).sum().item()
    cols = fl.sum(axis=0)
    print(f"frame {f}: flagged tiles {fl.sum()} of {fl.size}; short-history pixels {short}; flagged per tile column: "
          f"max {cols.max()} at column {cols.argmax()}, columns with any {int((cols > 0).sum())}; per tile row max {fl.sum(axis=1).max()}")
