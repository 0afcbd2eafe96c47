#!/usr/bin/env python3
"""What one rank of an N-GPU row-strip run of the 8K frame costs WITHOUT its halo exchanges: the strip
of rank R (4320/N rows + the redundant rows later passes tap) of the 7680x4320 frame of bench.py's
N > 1 workload (BASELINE configs[3]) on this one GPU, next to the unsharded 8K frame.  T(frame) /
T(strip) is the speed-up the redundant rows alone allow (strong scaling, upper bound); the exchanges
themselves are not in it (their bytes are printed; PROBE_LOOPBACK=1 times an RCCL send-to-self of the
same size on this GPU: the cost of the calls and of the copy through the device's own memory, not of a link).
    python3 tools/strip_probe.py [N [R]]        R defaults to rank 0 and an interior rank (N // 2)
    PROBE_EXCHANGE="-1 3 2"                     rmd_svgf_params.exchange_iteration values to compare (default "-1 3")"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402
from raymarchdenoisercuda_amd import sharding  # noqa: E402

W, H = 7680, 4320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RANKS = [int(sys.argv[2])] if len(sys.argv) > 2 else sorted({0, N // 2})
FRAMES, WARM = 24, 6
p = rmd.default_params()
p.max_motion_rows = 8


def run(den, frames, sync):
    for f in range(WARM):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(WARM, WARM + FRAMES):
        den(*frames[f % len(frames)])
    sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / FRAMES


d = rmd.SvgfDenoiser(W, H, params=p)
seq = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(8)]
ms1 = run(d.denoise, seq, lambda: None)
del d, seq
torch.cuda.empty_cache()
print(f"7680x4320 frame, one GPU      : {ms1:.4f} ms/frame  {W * H / ms1 / 1e3:.0f} Mpix/s")
for X in [int(v) for v in os.environ.get("PROBE_EXCHANGE", "-1 3").split()]:
    p.exchange_iteration = X
    for R in RANKS:
        sd = sharding.ShardedDenoiser(W, H, params=p, rank=R, world=N, pipelined=os.environ.get("PROBE_PIPELINE", "0") == "1",
                                      timing_only_no_exchange=True)
        seq = [sd.synth(f) for f in range(WARM + FRAMES)]
        msn = run(sd.denoise, seq, sd.synchronize)
        pl = sd.plan
        print(f"exchange_iteration {X:2d} rank {R} of {N}: rows [{pl.row0}, {pl.row1}) + buffer [{pl.buf_row0}, {pl.buf_row0 + pl.buf_rows}): "
              f"{msn:.4f} ms/frame -> speed-up bound {ms1 / msn:.2f}x of {N} (efficiency {ms1 / msn / N:.3f}); "
              f"received per frame: history {sharding.halo_bytes(pl, W) / 1e6:.2f} MB, mid-frame {sharding.mid_halo_bytes(pl, W) / 1e6:.2f} MB")
        del sd, seq
        torch.cuda.empty_cache()

if os.environ.get("PROBE_LOOPBACK") == "1":
    import ctypes as C
    from raymarchdenoisercuda_amd._lib import HaloStep, lib
    comm = C.c_void_p()
    rmd.check(lib.rmd_comm_create_all(1, None, C.byref(comm)))
    rows = 32
    plane = torch.rand((4 * rows, W, 4), device="cuda")
    steps = (HaloStep * 2)(HaloStep(HaloStep.RECV, HaloStep.PLANE_HIST_COLOR, 2 * rows, 3 * rows, 0), HaloStep(HaloStep.SEND, HaloStep.PLANE_HIST_COLOR, 0, rows, 0))
    stream = torch.cuda.current_stream().cuda_stream

    def once():
        rmd.check(lib.rmd_halo_exchange_steps(comm, 0, steps, 2, 0, 4 * rows, W, plane.data_ptr(), None, None, stream))
    for _ in range(5):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        once()
    e1.record()
    torch.cuda.synchronize()
    assert torch.equal(plane[2 * rows:3 * rows], plane[:rows])
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"RCCL loop-back (ncclSend + ncclRecv to self in one group) of {rows} rows x {W} px x 16 B = {rows * W * 16 / 1e6:.2f} MB: "
          f"{us:.1f} us per exchange = {rows * W * 16 / us / 1e3:.1f} GB/s through this GPU's memory (no xGMI link involved)")
    rmd.check(lib.rmd_comm_destroy(comm))
