#!/usr/bin/env python3
"""How much V work the 4K animated Cornell sequence (bench.py cornell_u8_sequence) asks for: per frame the 64x4 tiles T flags for
the spatial variance estimate, the pixels with a short history, the disocclusions.  (Steady state: ~2 500 of 32 400 tiles = 7.7 % against
2 % on the synthetic scene -- every silhouette of the tiled boxes loses its history under the fractional pan; that is the 197 against 169 us of
T+V on this sequence, profiles/r04_gbuffer_frame_gaps.txt.)"""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
import raymarchdenoisercuda_amd as rmd
import bench
W, H = 3840, 2160
pan = (2.25, 1.5)
p = rmd.default_params(); p.max_motion_rows = 8
seq = bench.cornell_u8_sequence(torch, W, H, 12, pan)
motion = torch.empty((H, W, 2), dtype=torch.float32, device="cuda"); motion[..., 0], motion[..., 1] = -pan[0], -pan[1]
den = rmd.SvgfDenoiser(W, H, params=p, debug=True)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
keep = []
for f, (rn, al, nm) in enumerate(seq):
    c = rmd.svgf.convert_u8_to_f32(rn, False, 0.0); a = rmd.svgf.convert_u8_to_f32(al, False, 0.0)
    rmd.svgf.demodulate(c, a, 1/255, out=c)
    nd = rmd.svgf.convert_u8_to_f32(nm, True, -1.0); keep.append(nd)
    den.denoise(c, nd, motion, out)
    torch.cuda.synchronize()
    h = den.t_debug[..., 3]
    print(f, "flagged tiles", int(den.tile_flags.sum().item()), "of", den.tile_flags.numel(), "pixels with h<4:", int((h < 4).sum().item()), "h==1:", int((h == 1).sum().item()), "mask!=15:", int((den.t_debug[..., 2] != 15).sum().item()))
