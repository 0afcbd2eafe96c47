#!/usr/bin/env python3
"""Does running the five a-trous iterations band by band (so that a band's planes stay in the 256 MB Infinity
Cache between iterations) beat five whole-frame launches?  Each band computes iteration i on the rows its later
iterations tap (redundant rows at the band borders); results are compared bit for bit with the whole-frame run.
    python3 tools/band_probe.py [bands ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
bands_list = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(3)]
for c, nd, m in frames:
    den.denoise(c, nd, m)
torch.cuda.synchronize()
c, nd, m = frames[-1]
desc = den.describe(c, nd, m, den.ping[1])
reach = [60, 56, 48, 32, 0]
bufs = [torch.empty_like(c) for _ in range(3)]


def run(nb):
    out = bufs[2]
    for b in range(nb):
        r0, r1 = H * b // nb, H * (b + 1) // nb
        src = den.v_color
        for it in range(5):
            dst = out if it == 4 else bufs[it & 1]
            a0, a1 = max(0, r0 - reach[it]), min(H, r1 + reach[it])
            rmd.svgf.atrous(desc, p, it, src, dst, a0, a1)
            src = dst
    return out


want = run(1).clone()
for nb in bands_list:
    got = run(nb)
    torch.cuda.synchronize()
    same = torch.equal(got, want)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        run(nb)
    e1.record()
    torch.cuda.synchronize()
    print(f"{nb} band(s): {e0.elapsed_time(e1) / reps * 1e3:8.1f} us for 5 iterations of {W}x{H}   bit-identical to one band: {same}")
