#!/usr/bin/env python3
"""Per-workgroup timeline of one a-trous launch (tuning aid, needs the trace build):
    tools/build_variant.sh trace -DRMD_ATROUS_TRACE
    RMD_LIB_PATH=build/variants/librmd_trace.so python3 tools/atrous_trace.py
Prints, per iteration, how the workgroups spread over XCDs / CUs, when they start and end
(100 MHz ticks = 10 ns), and the duration split of interior vs frame-edge workgroups."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import raymarchdenoisercuda_amd as rmd  # noqa: E402

W, H = (int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160)))
raw = C.CDLL(rmd.LIB_PATH)
p = rmd.default_params()
p.max_motion_rows = 8
p.atrous_variant = int(os.environ.get("PROBE_VARIANT", "0"))
den = rmd.SvgfDenoiser(W, H, params=p)
frames = [rmd.svgf.synth_gbuffer(W, H, f) for f in range(3)]
for c, nd, m in frames:
    den.denoise(c, nd, m)
torch.cuda.synchronize()
c, nd, m = frames[-1]
desc = den.describe(c, nd, m, den.ping[1])
NWG = 8192
PH = os.environ.get("RMD_TRACE_PHASES") is not None
buf = np.zeros(NWG * (14 if PH else 6), np.uint64)
src, dst = den.v_color, den.ping[0]
for it in range(5):
    for rep in range(2):                      # warm, then the launch that is read back (read clears)
        rmd.svgf.atrous(desc, p, it, src, dst, 0, H)
        assert raw.rmd_debug_atrous_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
    src, dst = dst, (den.ping[1] if dst is den.ping[0] else den.ping[0])
    rec = buf[:NWG * 6].reshape(NWG, 6)
    if PH:
        phs = buf[NWG * 6:].reshape(NWG, 8)[rec[:, 0] > 0][:, :5].astype(np.float64)
        tot = phs.sum(axis=1, keepdims=True)
        frac = np.median(phs / np.maximum(tot, 1), axis=0)
        if p.atrous_variant == 7:      # loader/consumer kernel: cycles, not fractions
            ph8 = buf[NWG * 6:].reshape(NWG, 8)[rec[:, 0] > 0].astype(np.float64)
            rows8 = np.maximum((rec[rec[:, 0] > 0][:, 5] & 0xFFFFFFFF).astype(np.float64), 1)
            med = np.median(ph8 / rows8[:, None], axis=0) * 12
            print("    cycles per step (12 lattice rows), median: consumer 0 wait %.0f compute %.0f loads+stores %.0f | consumer 11 wait %.0f compute %.0f"
                  " | loader wait-for-slot %.0f commit %.0f issue %.0f" % tuple(med))
        else:
            print("    wave-0 time split (median over workgroups): issue loads %.3f  compute %.3f  barrier1 %.3f  store %.3f  barrier2 %.3f" % tuple(frac))
    t = rec[rec[:, 0] > 0]
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0).astype(np.int64), (t[:, 1] - t0).astype(np.int64)
    dur = end - start
    hw, xcc = (t[:, 2] & 0xFFFFFFFF).astype(np.int64), (t[:, 2] >> 32).astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    edge = (t[:, 3] & 1).astype(bool)
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
    print(f"--- iteration {it} (step {1 << it}): {len(t)} workgroups, kernel span {end.max() / 100:.1f} us")
    print(f"    distinct CUs {len(per_cu)}, workgroups per CU min/max {per_cu.min()}/{per_cu.max()}, histogram {np.bincount(per_cu).tolist()}")
    print(f"    per XCD: {np.bincount(xcc, minlength=8).tolist()}")
    print(f"    start  us: p50 {np.percentile(start, 50) / 100:.1f}  p90 {np.percentile(start, 90) / 100:.1f}  max {start.max() / 100:.1f}")
    print(f"    end    us: p10 {np.percentile(end, 10) / 100:.1f}  p50 {np.percentile(end, 50) / 100:.1f}  p90 {np.percentile(end, 90) / 100:.1f}  max {end.max() / 100:.1f}")
    for name, sel in (("interior", ~edge), ("edge", edge)):
        if sel.any():
            d = dur[sel] / 100.0
            print(f"    {name:8s} n={sel.sum():4d} duration us: min {d.min():.1f}  p50 {np.percentile(d, 50):.1f}  p90 {np.percentile(d, 90):.1f}  max {d.max():.1f}")
    print("    per-XCD end us (p50/max): " + "  ".join(f"{np.percentile(end[xcc == k], 50) / 100:.0f}/{end[xcc == k].max() / 100:.0f}" for k in range(8)))
    cu_end = np.array([end[cu_key == k].max() for k in np.unique(cu_key)]) / 100.0
    print(f"    per-CU last end us: p10 {np.percentile(cu_end, 10):.1f}  p50 {np.percentile(cu_end, 50):.1f}  p90 {np.percentile(cu_end, 90):.1f}  max {cu_end.max():.1f}")
    late = start > np.percentile(end, 10)
    print(f"    workgroups that start after the first 10% have ended: {late.sum()}")
    cyc, rows = t[:, 4].astype(np.float64), (t[:, 5] & 0xFFFFFFFF).astype(np.float64)
    ghz = cyc / np.maximum(dur, 1) * 0.1
    sel = ~edge
    print(f"    shader clock while resident: p50 {np.percentile(ghz, 50):.2f} GHz (p10 {np.percentile(ghz, 10):.2f}, p90 {np.percentile(ghz, 90):.2f}); "
          f"interior workgroups: {np.percentile(cyc[sel] / rows[sel], 50):.0f} cycles per lattice row of 128 px (p90 {np.percentile(cyc[sel] / rows[sel], 90):.0f})")
    # placement: is a workgroup's "layer" on its CU (how many workgroups with a smaller id share the CU) the dispatch order
    # within its XCD?  (pid = xcd + 8 k: the k-th workgroup of that XCD)
    pid = np.nonzero(rec[:, 0] > 0)[0]
    kk = pid >> 3
    layer = np.zeros(len(pid), np.int64)
    for key in np.unique(cu_key):
        idx = np.nonzero(cu_key == key)[0]
        layer[idx[np.argsort(pid[idx])]] = np.arange(len(idx))
    print("    placement: k = pid >> 3 of the workgroups that are the 1st / 2nd / 3rd on their CU: " + "  ".join(
        f"layer {q}: n {np.sum(layer == q)} k p5 {np.percentile(kk[layer == q], 5):.0f} p50 {np.percentile(kk[layer == q], 50):.0f} p95 {np.percentile(kk[layer == q], 95):.0f}"
        for q in range(int(layer.max()) + 1)) + f";  XCC_ID == pid & 7 for {np.mean(xcc == (pid & 7)) * 100:.0f} % of them")
    occ = (dur.sum() / max(end.max(), 1)) / (len(per_cu) * (1 if p.atrous_variant == 7 else 3))
    print(f"    average resident workgroups / (CUs x slots per CU): {occ:.3f}")
