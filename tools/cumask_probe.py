#!/usr/bin/env python3
"""What a CU-partition stream (rmd_stream_create_partition) gets: bandwidth of a plain copy, the T+V half of a
frame and the five a-trous launches, each alone on its side of the split.
    python3 tools/cumask_probe.py [reserve_per_xcd ...]      default 4 8 16"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
from raymarchdenoisercuda_amd._lib import SvgfParams  # noqa: E402

lib = rmd.lib
W, H = 3840, 2160
p = rmd.default_params()
p.max_motion_rows = 8
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(4)]
for c, nd, m in frames[:3]:
    den.denoise(c, nd, m)
torch.cuda.synchronize()
c, nd, m = frames[3]
out = torch.empty_like(c)
desc = den.describe(c, nd, m, out)
src = torch.empty((256 << 20,), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)


def timed(stream, fn, reps=10):
    with torch.cuda.stream(stream):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run_tv(stream):
    rmd.check(lib.rmd_svgf_frame_tv(C.byref(desc), C.byref(p), 0, H, stream.cuda_stream))


def run_a(stream, params):
    s, d = den.v_color, den.ping[0]
    for it in range(5):
        rmd.svgf.atrous(desc, params, it, s, d, 0, H, stream=stream)
        s, d = d, (den.ping[1] if d is den.ping[0] else den.ping[0])


full = torch.cuda.Stream()
print(f"all 256 CUs: copy 256 MiB {timed(full, lambda: dst.copy_(src)):8.1f} us   T+V {timed(full, lambda: run_tv(full)):8.1f} us   "
      f"a-trous x5 {timed(full, lambda: run_a(full, p)):8.1f} us")
keep = []
for r in [int(v) for v in sys.argv[1:]] or [4, 8, 16]:
    raw, n = [C.c_void_p(), C.c_void_p()], [C.c_int(), C.c_int()]
    for side in (0, 1):
        rmd.check(lib.rmd_stream_create_partition(C.byref(raw[side]), r, side, C.byref(n[side])))
    keep.append(raw)
    sa, sb = (torch.cuda.ExternalStream(x.value) for x in raw)
    pa = SvgfParams.from_buffer_copy(p)
    pa.atrous_cus = n[0].value
    print(f"reserve {r:2d} per XCD: side 1 ({n[1].value:3d} CUs) copy {timed(sb, lambda: dst.copy_(src)):8.1f} us   T+V {timed(sb, lambda: run_tv(sb)):8.1f} us   "
          f"| side 0 ({n[0].value:3d} CUs) a-trous x5 {timed(sa, lambda: run_a(sa, pa)):8.1f} us, planned for 256 CUs {timed(sa, lambda: run_a(sa, p)):8.1f} us", flush=True)
torch.cuda.synchronize()
os._exit(0)      # the partition streams are left to the process exit
