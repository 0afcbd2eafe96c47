#!/usr/bin/env python3
"""T+V (rmd_svgf_frame_tv) timed with HIP events in frame context -- the launch runs between the a-trous iterations of
consecutive frames, as in the frame loop -- median over PROBE_FRAMES frames.  For A/B runs of the kernel's knobs with the
experiments build (RMD_TV_XCD_GROUP=0|1|2|4|8: how the interior tiles are dealt out to the XCDs).
    PROBE_W / PROBE_H (3840 x 2160), PROBE_FRAMES (60)"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
FRAMES = int(os.environ.get("PROBE_FRAMES", 60))
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
seq = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(24)]
out = torch.empty_like(seq[0][0])
for k in range(150):                                   # history + clock preconditioning
    den.denoise(*seq[k % 24], out)
torch.cuda.synchronize()
timer = C.c_void_p()
rmd.check(rmd.lib.rmd_timer_create(C.byref(timer)))
ms, times = C.c_float(), []
for k in range(FRAMES):
    c, nd, m = seq[(150 + k) % 24]
    d = den.describe(c, nd, m, out)
    rmd.check(rmd.lib.rmd_timer_start(timer, None))
    rmd.check(rmd.lib.rmd_svgf_frame_tv(C.byref(d), C.byref(p), 0, H, None))
    rmd.check(rmd.lib.rmd_timer_stop(timer, None))
    rmd.check(rmd.lib.rmd_svgf_frame_atrous(C.byref(d), C.byref(p), 0, H, None, None))
    rmd.check(rmd.lib.rmd_timer_elapsed_ms(timer, C.byref(ms)))
    times.append(ms.value * 1e3)
    den.cur ^= 1
    den.has_history, den.prev_nd = True, nd
times.sort()
print(f"T+V {W}x{H} RMD_TV_XCD_GROUP={os.environ.get('RMD_TV_XCD_GROUP', 'default')}: median {statistics.median(times):.1f} us, "
      f"p10 {times[len(times) // 10]:.1f}, p90 {times[len(times) * 9 // 10]:.1f}, flagged tiles {int(den.tile_flags.sum().item())}")
