#!/usr/bin/env python3
"""bench.py — full SVGF (temporal + variance + 5 a-trous) throughput on synthetic G-buffers.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one frame: T + V (one launch) + 5 x A over the whole frame (6 kernel launches), inputs resident in
HBM before the timed region (all W+K frames of G-buffer are pre-generated on the device).
  N = 1 : BASELINE.json configs[2], 3840x2160 synthetic G-buffer + radiance, fp32.
  N > 1 : BASELINE.json configs[3], the fixed 7680x4320 (8K) frame cut into N row strips of 4320/N
          rows, one per GPU (STRONG scaling); ranks exchange the history halo rows with rank+-1
          over RCCL every frame, plus ONE exchange inside the frame (a-trous iteration 3's halo rows), sharding.py.  Rank 0 also times the unsharded 8K frame on its
          own GPU after the timed region (`one_gpu_same_frame`), so the speed-up is on one workload.
Without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the parent starts
the N ranks itself as child processes BEFORE anything touches the GPU, relays rank 0's JSON line
and exits non-zero if any rank fails.
Rank 0 prints ONE JSON line.  `value` = Mpixels/s of the whole job.  `roofline` is measured live
with HIP events around single launches of the dominant kernel (the a-trous iteration: 48 B/px
algorithmic = 32 read + 16 written).  `cpu_baseline` times the scalar oracle
(oracle/, a port: the reference has no CPU path) on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
ATROUS_BYTES_PER_PX = 48       # per iteration: color 16 + nd 16 read, color 16 written
TV_BYTES_PER_PX = 106          # T + V in one launch: 81 read + 25 written (measure_roofline)
FULL_BYTES_PER_PX = 424        # T 120 + V 64 + 5 x 48, every pass priced on its own
MOVED_BYTES_PER_PX = 346       # what rmd_svgf_frame moves: T+V 81 read (color 16, nd 16, motion 8, prev_nd 16, hist_color 16, hist_moments 8,
                               # hist_len 1) + 25 written (v_color 16, t_moments 8, t_len 1; V's windows are recomputed in the T
                               # workgroup, no t_color plane), 5 x 48
MAX_RESIDENT = 64              # pre-generated G-buffer frames kept in HBM


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-sizes", action="store_true")
    ap.add_argument("--roofline-reps", type=int, default=30)
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="start the ranks, rendezvous, reduce one number, print a JSON line; no GPU work (CPU test of the launcher)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher: start N ranks as fresh child processes (one per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and relay rank 0's output.
    Runs before torch or librmd are imported: the parent never touches the GPU and nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None, text=True))
    import threading

    def relay():                            # rank 0 prints the one JSON line; everything else goes to stderr
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    rc, deadline = 0, time.time() + float(os.environ.get("RMD_BENCH_LAUNCH_TIMEOUT", "900"))
    while any(pr.poll() is None for pr in procs):
        failed = [pr.returncode for pr in procs if pr.poll() not in (None, 0)]
        if failed or time.time() > deadline:
            rc = (failed[0] if failed and failed[0] > 0 else 1)
            for pr in procs:                # a rank died: the others would wait in a collective forever
                if pr.poll() is None:
                    pr.terminate()
            break
        time.sleep(0.1)
    for pr in procs:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
        if pr.returncode != 0 and rc == 0:
            rc = pr.returncode if pr.returncode > 0 else 1
    reader.join(timeout=5)
    return rc


def rehearse_launcher(args):
    """The ranks' side of --rehearse-launcher: rendezvous over gloo, one MAX all-reduce, one JSON line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if os.environ.get("RMD_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_rehearsal": True, "n_gpus": world, "max_over_ranks": float(t.item())}), flush=True)
    dist.destroy_process_group()


def hip_event_ms(rmd, fn, reps):
    """Average duration of fn() in ms, each call bracketed by HIP events on the launch stream (NULL)."""
    timer = C.c_void_p()
    rmd.check(rmd.lib.rmd_timer_create(C.byref(timer)))
    times = []
    ms = C.c_float()
    for _ in range(reps):
        rmd.check(rmd.lib.rmd_timer_start(timer, None))
        fn()
        rmd.check(rmd.lib.rmd_timer_stop(timer, None))
        rmd.check(rmd.lib.rmd_timer_elapsed_ms(timer, C.byref(ms)))
        times.append(ms.value)
    rmd.lib.rmd_timer_destroy(timer)
    return times


def measure_roofline(rmd, torch, den, frames, width, rows_out, plan, reps, whole_frame_4k):
    """Dominant kernel = the a-trous iteration.  Each of the 5 launches of a frame is timed with HIP events on
    the stream it runs on, IN THE CONTEXT of a frame: per repetition T+V run first (rmd_svgf_frame_tv), then the
    iterations with the plane routing of rmd_svgf_frame -- the cache state a launch meets in the pipeline, so
    the figures agree with the per-kernel averages of a rocprofv3 trace of the frame loop.  (Launches repeated
    in isolation on unchanged planes run 5-7 % faster since the outputs are stored non-temporally.)"""
    p = den.params
    row0, row1 = plan.row0, plan.row1
    n = p.iterations
    timer = C.c_void_p()
    rmd.check(rmd.lib.rmd_timer_create(C.byref(timer)))
    ms = C.c_float()
    out = torch.empty_like(frames[0][0])
    it_reach = rmd.svgf.frame_iteration_reach(p)     # rows beyond the strip each iteration is computed on (rmd_svgf_frame)
    H = den.height
    sums, launch_px = [0.0] * n, [0] * n
    tv_ms = []
    for rep in range(reps + 2):
        color, nd, motion = frames[rep % len(frames)]
        desc = den.describe(color, nd, motion, out)
        rmd.check(rmd.lib.rmd_timer_start(timer, None))
        rmd.check(rmd.lib.rmd_svgf_frame_tv(C.byref(desc), C.byref(p), row0, row1, None))
        rmd.check(rmd.lib.rmd_timer_stop(timer, None))
        rmd.check(rmd.lib.rmd_timer_elapsed_ms(timer, C.byref(ms)))
        if rep >= 2:
            tv_ms.append(ms.value)
        src, pp = den.v_color, 0
        for it in range(n):
            if it == n - 1:
                dst = out
            elif it == p.hist_iteration:
                dst = den.hist_color[den.cur ^ 1]
            else:
                dst, pp = den.ping[pp], pp ^ 1
            a0, a1 = max(0, row0 - it_reach[it]), min(H, row1 + it_reach[it])
            rmd.check(rmd.lib.rmd_timer_start(timer, None))
            rmd.svgf.atrous(desc, p, it, src, dst, a0, a1)
            rmd.check(rmd.lib.rmd_timer_stop(timer, None))
            rmd.check(rmd.lib.rmd_timer_elapsed_ms(timer, C.byref(ms)))
            if rep >= 2:
                sums[it] += ms.value
            launch_px[it] = width * (a1 - a0)                        # a strip's launches also produce the rows later iterations tap
            src = dst
        den.cur ^= 1
        den.has_history, den.prev_nd = True, nd
    rmd.lib.rmd_timer_destroy(timer)
    per_iter = [v / reps for v in sums]
    px = statistics.mean(launch_px)
    avg_ms = statistics.mean(per_iter)
    achieved = ATROUS_BYTES_PER_PX * px / (avg_ms * 1e-3) / 1e9          # bytes of the average launch / the average launch time
    # PMC-measured HBM bytes per launch, if a rocprofv3 --pmc pass of this command was reduced
    # into profiles/ (tools/pmc_traffic.py); null otherwise.
    traffic, valu, tv_traffic = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    # (the counter passes were taken on whole 3840x2160 launches: they say nothing about a strip's launches)
    if whole_frame_4k and os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath))
            traffic = pmc.get("atrous_hbm_bytes_per_launch")
            tv_traffic = (pmc.get("temporal_variance") or {}).get("hbm_bytes_per_launch")
            vi = pmc.get("valu_issue")
            if vi:   # the bound the kernel actually sits on (PMC passes of the same kernels, tools/pmc_passes.sh)
                valu = {"busy_frac": vi["avg_valu_busy_frac"], "resident_waves_per_simd": vi["avg_resident_waves_per_simd"],
                        "max_waves_per_simd": 3, "source": "profiles/pmc_traffic.json (rocprofv3 --pmc)"}
        except Exception:
            traffic, valu, tv_traffic = None, None, None
    return {
        "bound": "hbm", "kernel": "atrous_stream_kernel<S,2> (one a-trous iteration, avg over S=1,2,4,8,16)",
        "note": "priced against HBM as BASELINE.json asks; the kernel is VALU-issue bound (DESIGN.md §4)",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "valu_issue": valu,
        "algorithmic_bytes_per_launch": int(ATROUS_BYTES_PER_PX * px),
        "timing": "HIP events around each launch, in frame context (T+V, then the 5 iterations with rmd_svgf_frame's routing)",
        "avg_launch_ms": round(avg_ms, 5),
        "per_iteration_ms": [round(v, 5) for v in per_iter],
        "atrous_x5_ms": round(sum(per_iter), 5),
        "atrous_x5_mpix_s": round(width * rows_out / (sum(per_iter) * 1e-3) / 1e6, 1),
        # the frame's other kernel, HBM-bound: T + V in one launch, 106 B/px algorithmic (81 read: color 16, nd 16, motion 8, prev_nd 16,
        # hist_color 16, hist_moments 8, hist_len 1; 25 written: v_color 16, t_moments 8, t_len 1); MEDIAN launch (the frames that
        # restart the history run V everywhere), same timing method; rows = the strip's T rows on N > 1
        "temporal_variance": tv_roofline(tv_ms, width * (min(H, row1 + rmd.svgf.frame_reach(p)[3]) - max(0, row0 - rmd.svgf.frame_reach(p)[3])), tv_traffic),
    }


def tv_roofline(tv_ms, px, traffic):
    med = statistics.median(tv_ms)
    achieved = TV_BYTES_PER_PX * px / (med * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "svgf_temporal_variance_kernel (T + V, one launch)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": int(TV_BYTES_PER_PX * px),
            "median_launch_ms": round(med, 5)}


def other_size(rmd, torch, width, height, p, frames=16, warm=4):
    """Full SVGF Mpixels/s at another frame size: `warm` + `frames` frames of the synthetic sequence."""
    den = rmd.SvgfDenoiser(width, height, params=p)
    seq = [rmd.svgf.synth_gbuffer(width, height, f) for f in range(warm + frames)]
    out = torch.empty_like(seq[0][0])
    t_pre = time.perf_counter()                      # clock preconditioning as in main(): ~60 ms of the same loop, then a fresh history
    while time.perf_counter() - t_pre < 0.06:
        for f in range(warm + frames):
            den.denoise(*seq[f], out)
        torch.cuda.synchronize()
    den.reset_history()
    for f in range(warm):
        den.denoise(*seq[f], out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(warm, warm + frames):
        den.denoise(*seq[f], out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"mpix_s": round(width * height * frames / dt / 1e6, 1), "ms_per_frame": round(dt / frames * 1e3, 4),
            "effective_GBps": round(FULL_BYTES_PER_PX * width * height * frames / dt / 1e9, 1),
            "moved_GBps": round(MOVED_BYTES_PER_PX * width * height * frames / dt / 1e9, 1), "frames": frames}


def cornell_u8_sequence(torch, width, height, nframes, pan, seed=2024):
    """uint8 G-buffer frames of the animated Cornell sequence (BASELINE configs[4]), resident on the device: the 500x500
    fixture planes (tests/golden/cornell = the reference's render/cornell/1) tiled to the frame and moved by `pan` pixels per
    frame.  The pan is FRACTIONAL: frame f is the tiled picture resampled bilinearly at x - f * pan (the tiled picture is
    periodic, so this is a camera pan with no border), so that motion = -pan is physically right and the reprojection's
    bilinear weights are non-trivial.  Per-frame re-seeded noise multiplies the radiance."""
    import numpy as np
    from PIL import Image
    gold = os.path.join(ROOT, "tests", "golden", "cornell")

    def plane(name):
        rgb = np.array(Image.open(os.path.join(gold, f"{name}.png")).convert("RGB"))
        rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
        t = torch.from_numpy(np.ascontiguousarray(rgba)).cuda().float()
        reps = (-(-height // t.shape[0]), -(-width // t.shape[1]), 1)
        return t.repeat(*reps)[:height, :width].contiguous()

    base = {n: plane(n) for n in ("render", "albedo", "normal")}
    g = torch.Generator(device="cuda").manual_seed(seed)

    def moved(t, f):
        ox, oy = f * pan[0], f * pan[1]
        ix, iy = int(ox // 1), int(oy // 1)
        fx, fy = ox - ix, oy - iy
        r = lambda dx, dy: torch.roll(t, shifts=(iy + dy, ix + dx), dims=(0, 1))          # noqa: E731
        # content(x) = base(x - offset): bilinear in the two neighbours on either axis
        return (1 - fx) * (1 - fy) * r(0, 0) + fx * (1 - fy) * r(1, 0) + (1 - fx) * fy * r(0, 1) + fx * fy * r(1, 1)

    seq = []
    for f in range(nframes):
        render = moved(base["render"], f)
        render[..., :3] *= 0.75 + 0.5 * torch.rand((height, width, 1), device="cuda", generator=g)
        u8 = lambda t: t.add(0.5).clamp_(0, 255).to(torch.uint8).contiguous()                # noqa: E731
        rn, al, nm = u8(render), u8(moved(base["albedo"], f)), u8(moved(base["normal"], f))
        rn[..., 3], al[..., 3], nm[..., 3] = 255, 255, 255                                   # opaque alpha: depth 1 (include/rmd_api.h)
        seq.append((rn, al, nm))
    return seq


def pcie_streaming(rmd, torch, gden, seq, motion, frames, warm, host_frames=8):
    """uchar4 planes in pinned HOST memory -> H2D -> rmd_svgf_gbuffer_frame -> D2H of `denoised`, software-pipelined: two device
    GBuffers, the uploads on one stream, the frames on torch's current stream, the downloads on a third, ordered by events.  In
    the steady state a frame costs max(upload of 12 B/px, the frame, download of 4 B/px); at 4K the upload (100 MB per frame
    over the host link) is the longest of the three."""
    host = [tuple(t.cpu().pin_memory() for t in fr) for fr in seq[:host_frames]]
    h, w = seq[0][0].shape[:2]
    dev_in = [tuple(torch.empty_like(seq[0][0]) for _ in range(3)) for _ in range(2)]
    dev_out = [torch.empty_like(seq[0][0]) for _ in range(2)]
    host_out = [torch.empty(seq[0][0].shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
    main, up, down = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev_up, ev_done, ev_down = ([torch.cuda.Event() for _ in range(2)] for _ in range(3))
    gden.reset_history()

    def frame(f):
        k = f & 1
        with torch.cuda.stream(up):
            up.wait_event(ev_done[k])                    # the frame that last read these device planes (f - 2) is done
            for dst, src in zip(dev_in[k], host[f % host_frames]):
                dst.copy_(src, non_blocking=True)
            ev_up[k].record(up)
        main.wait_event(ev_up[k])
        main.wait_event(ev_down[k])                      # the download of frame f - 2 has left dev_out[k]
        gden.frame(dev_in[k][0], dev_in[k][1], dev_in[k][2], dev_out[k], motion)
        ev_done[k].record(main)
        with torch.cuda.stream(down):
            down.wait_event(ev_done[k])
            host_out[k].copy_(dev_out[k], non_blocking=True)
            ev_down[k].record(down)

    for f in range(warm):
        frame(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(warm, warm + frames):
        frame(f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    px = w * h
    return {"fps": round(frames / dt, 1), "ms_per_frame": round(dt / frames * 1e3, 4), "mpix_s": round(px * frames / dt / 1e6, 1),
            "h2d_GBps": round(12.0 * px * frames / dt / 1e9, 1), "d2h_GBps": round(4.0 * px * frames / dt / 1e9, 1),
            "what": "pinned host uchar4 planes -> H2D (12 B/px) -> one-call frame -> D2H of denoised (4 B/px); 3 streams, 2 device GBuffers",
            "last_frame_mean_u8": round(float(host_out[(warm + frames - 1) & 1][..., :3].float().mean()), 2)}


def cornell_sequence(rmd, torch, p, width=3840, height=2160, frames=60, warm=6, pan=(2.25, 1.5)):
    """BASELINE configs[4]: a 60-frame 4K animated Cornell sequence, steady-state frames per second, in three forms over the
    SAME resident frames:
      end_to_end_u8.fused    uchar4 render / albedo / normal in -> uchar4 denoised out, ONE call per frame on the reference's
                             GBuffer (rmd_svgf_gbuffer_frame: 6 launches, the 8-bit ends inside the first and the last)
      end_to_end_u8.unfused  the same bytes in and out through the eight-call chain (3 x rmd_convert_u8_to_f32, rmd_demodulate,
                             the float-plane frame, rmd_convert_f32_to_u8: 11 launches)
      float_planes           float planes in, float plane out (conversions and demodulation done before the clock, modulation
                             after it): what round 3 reported as this configuration's fps
    The last frame of the two 8-bit forms is compared byte for byte."""
    eps = 1.0 / 255.0
    seq = cornell_u8_sequence(torch, width, height, warm + frames, pan)
    motion = torch.empty((height, width, 2), dtype=torch.float32, device="cuda")
    motion[..., 0], motion[..., 1] = -float(pan[0]), -float(pan[1])
    n = warm + frames

    def clock(step, reset):
        t_pre = time.perf_counter()                  # clock preconditioning as in main(), then a fresh history
        while time.perf_counter() - t_pre < 0.06:
            for f in range(n):
                step(f)
            torch.cuda.synchronize()
        reset()
        for f in range(warm):
            step(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(warm, n):
            step(f)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"fps": round(frames / dt, 1), "ms_per_frame": round(dt / frames * 1e3, 4),
                "mpix_s": round(width * height * frames / dt / 1e6, 1)}

    # --- fused: one call per frame on the GBuffer
    gden = rmd.GBufferDenoiser(width, height, params=p, albedo_eps=eps)
    out_fused = torch.empty_like(seq[0][0])
    fused = clock(lambda f: gden.frame(seq[f][0], seq[f][1], seq[f][2], out_fused, motion), gden.reset_history)
    fused["launches_per_frame"] = 6
    # --- the same call with the frames coming from and going back to HOST memory (the CudaGBuffer::openImages direction of the
    # boundary, include/gbuffer.h:20-33): pinned host planes, double-buffered device GBuffers, uploads / the frame / the
    # download on three streams.  NOT `value` (inputs resident is the contract); the PCIe-inclusive rate DESIGN.md quotes.
    pcie = pcie_streaming(rmd, torch, gden, seq, motion, frames, warm)
    del gden
    # --- unfused: the eight-call chain on the same bytes
    den = rmd.SvgfDenoiser(width, height, params=p)
    color, alb = torch.empty((height, width, 4), dtype=torch.float32, device="cuda"), torch.empty((height, width, 4), dtype=torch.float32, device="cuda")
    nds = [torch.empty_like(color), torch.empty_like(color)]          # nd is borrowed as prev_nd until the next frame
    out_f32, out_chain = torch.empty_like(color), torch.empty_like(seq[0][0])
    npx = width * height

    def chain_step(f):
        rn, al, nm = seq[f]
        nd = nds[f & 1]
        rmd.check(rmd.lib.rmd_convert_u8_to_f32(rn.data_ptr(), color.data_ptr(), npx, 0, 0.0, None))
        rmd.check(rmd.lib.rmd_convert_u8_to_f32(al.data_ptr(), alb.data_ptr(), npx, 0, 0.0, None))
        rmd.check(rmd.lib.rmd_convert_u8_to_f32(nm.data_ptr(), nd.data_ptr(), npx, 1, -1.0, None))
        rmd.check(rmd.lib.rmd_demodulate(color.data_ptr(), alb.data_ptr(), color.data_ptr(), npx, eps, None))
        den.denoise(color, nd, motion, out_f32)
        rmd.check(rmd.lib.rmd_convert_f32_to_u8(out_f32.data_ptr(), alb.data_ptr(), out_chain.data_ptr(), npx, None))

    unfused = clock(chain_step, den.reset_history)
    unfused["launches_per_frame"] = 11
    same = bool(torch.equal(out_chain, out_fused))
    del den
    # --- float planes in / out (round 3's figure for this configuration): everything 8-bit outside the clock
    fseq = []
    for rn, al, nm in seq:
        c = rmd.svgf.convert_u8_to_f32(rn, False, 0.0)
        a = rmd.svgf.convert_u8_to_f32(al, False, 0.0)
        rmd.svgf.demodulate(c, a, eps, out=c)
        fseq.append((c, rmd.svgf.convert_u8_to_f32(nm, True, -1.0)))
        del a
    den = rmd.SvgfDenoiser(width, height, params=p)
    floats = clock(lambda f: den.denoise(fseq[f][0], fseq[f][1], motion, out_f32), den.reset_history)
    return {"workload": f"{frames}-frame {width}x{height} animated Cornell sequence (tiled fixture planes resampled at a fractional pan of "
                        f"{pan} px/frame, re-seeded noise), full SVGF fp32, all frames resident as uchar4 planes",
            "frames": frames,
            "end_to_end_u8": {"fused": fused, "unfused": unfused, "last_frame_bytes_identical": same,
                              "fused_over_unfused": round(fused["fps"] / unfused["fps"], 3)},
            "pcie_inclusive_u8": pcie,
            "float_planes": floats, "fused_u8_over_float_planes": round(fused["fps"] / floats["fps"], 3),
            # (kept under the old keys: the figure comparable with round 3's cornell_sequence_4k)
            "fps": fused["fps"], "ms_per_frame": fused["ms_per_frame"], "mpix_s": fused["mpix_s"],
            "last_frame_mean_u8": round(float(out_fused[..., :3].float().mean()), 2)}


def reference_api_kernels(rmd, torch, width=3840, height=2160):
    """The reference's own entry points (uchar4 box mean, radius 2, depth 1: src/test.cu:68-90) on a 4K
    plane: microseconds per launch and algorithmic GB/s (8 B/px).  Not the headline value."""
    g = torch.Generator(device="cuda").manual_seed(7)
    render = torch.randint(0, 256, (height, width, 4), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty_like(render)
    frame = rmd.make_gbuffer(render, out)
    p = rmd.FilterParams(radius=2, depth=1)
    res = {}
    for name, fn in (("filterKernelBaseline", rmd.filterKernelBaseline), ("filterKernelTiled", rmd.filterKernelTiled)):
        for _ in range(3):
            fn(frame, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(frame, p)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        res[name] = {"us_per_launch": round(us, 1), "GBps_algorithmic": round(8.0 * width * height / us / 1e3, 1)}
    res["workload"] = f"{width}x{height} uchar4, FilterParams{{AVERAGE, radius 2, depth 1}}"
    return res


def usable_cores(hardware_threads):
    """Host threads this process may actually run at once: affinity mask and cgroup CPU quota (the
    one-GPU boxes of this pool expose 256 hardware threads behind a 16-CPU quota; 256 oracle threads
    there run SLOWER than 16)."""
    n = min(hardware_threads, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline():
    """Scalar oracle on all usable host threads at the HEADLINE configuration (3840x2160 when a frame takes
    <= 30 s; it takes about a second): frames 1..2 of a 3-frame 3840x2160 synthetic sequence of full SVGF (frame 0 only builds
    history).  The 1080p and single-thread figures of the earlier rounds are kept under `other`.  A reported baseline, not a
    target; the reference has no CPU path to time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    cores = usable_cores(orc.hardware_threads())
    p = orc.default_params()

    def timed(w, h, frames, threads):
        hc = hm = pn = None
        dt = 0.0
        for f in range(frames):
            c, nd, m = orc.synth_gbuffer(w, h, f)
            fr = orc.Frame(w, h, c, nd, m, hc, hm, pn, debug=False)
            t0 = time.perf_counter()
            orc.frame(fr, p, threads=threads if f > 0 else cores)
            if f > 0:
                dt += time.perf_counter() - t0
            hc, hm, pn = fr.history()
        return (frames - 1) * w * h / dt / 1e6, dt

    v4k, dt4k = timed(3840, 2160, 3, cores)
    v1080, dt1080 = timed(1920, 1080, 3, cores)
    v1, dt1 = timed(480, 270, 2, 1)              # ONE thread (SURVEY §8d asks for both), 1/16 of 1080p
    return {"value": round(v4k, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"2 frames (after 1 history-building frame) of full SVGF at 3840x2160 synthetic (BASELINE configs[2], the headline "
                      f"configuration), scalar C oracle (gcc -O2 -ffp-contract=off), static row strips on the {cores} host threads the "
                      f"process may use ({orc.hardware_threads()} hardware threads visible)",
            "seconds": round(dt4k, 3), "core_seconds": round(dt4k * cores, 1),
            "other": {"1920x1080": {"value": round(v1080, 3), "unit": "Mpixels/s", "cores": cores, "sample": "2 frames with history", "seconds": round(dt1080, 3)},
                      "single_thread": {"value": round(v1, 4), "unit": "Mpixels/s", "cores": 1,
                                        "sample": "1 frame with history of full SVGF at 480x270 synthetic", "seconds": round(dt1, 3)}}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.rehearse_launcher:
        return rehearse_launcher(args)
    import torch
    import torch.distributed as dist
    import raymarchdenoisercuda_amd as rmd        # ImportError if librmd.so is missing: there is no fallback
    from raymarchdenoisercuda_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # RMD_DIST_BACKEND=gloo rehearses the N>1 path with several ranks sharing one GPU (RCCL refuses
    # two ranks on one device); the driver's runs use the default: one rank per GPU over RCCL/xGMI.
    backend = os.environ.get("RMD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    rmd.check(rmd.lib.rmd_set_device(local))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if world == 1:
        width, height = 3840, 2160
        workload = "3840x2160 synthetic G-buffer + radiance, full SVGF fp32 (BASELINE configs[2])"
    else:
        width, height = 7680, 4320
        workload = (f"7680x4320 (8K) synthetic G-buffer in {world} row strips of {height // world} rows, one per GPU "
                    "(BASELINE configs[3], strong scaling), full SVGF fp32, neighbour halo exchanges (history + one inside the frame) over "
                    + ("RCCL / xGMI" if backend == "nccl" else f"torch.distributed '{backend}' (a rehearsal on shared devices, not xGMI)"))
    p = rmd.default_params()
    p.max_motion_rows = 8            # the synthetic pan moves <= 1.5 rows per frame
    p.atrous_variant = int(os.environ.get("RMD_ATROUS_VARIANT", "0"))   # 0 = library default (experiments only)
    # Default (every N): the 6 launches of a frame back to back on one stream, so the per-kernel durations of a
    # rocprofv3 run of this command are those of isolated launches (what `roofline` prices).
    # RMD_PIPELINE=1 software-pipelines consecutive frames over two HIP streams (T+V of frame k+1
    # under the a-trous iterations of frame k): measured SLOWER on one GPU (round 3: 9260 against 9620 Mpix/s), so it is
    # not the default for any N; with N > 1 the exchanges have a stream of their own either way.
    pipelined = os.environ.get("RMD_PIPELINE", "0") == "1"
    # N > 1: ONE neighbour exchange inside the frame (a-trous iteration 3's 32 halo rows per side, 3.9 MB at 8K, beside the
    # interior rows of iteration 3) instead of redundant rows only: T, V, A0..A2 run on 32 fewer rows per side
    # (sharding.ShardedDenoiser; RMD_EXCHANGE_ITERATION=-1 switches it off).  The exchanges run on a second stream.
    if world > 1 and not pipelined:
        p.exchange_iteration = int(os.environ.get("RMD_EXCHANGE_ITERATION", "3"))
    sd = sharding.ShardedDenoiser(width, height, params=p, rank=rank, world=world, pipelined=pipelined)
    plan = sd.plan
    rows_out = plan.row1 - plan.row0

    # all frames resident in HBM before timing (40 B/px/frame: sized for 288 GB, not streamed).  At
    # most MAX_RESIDENT frames are kept (~21 GB at 4K); a longer run cycles through them, which costs
    # one reprojection miss (a burst of short-history pixels) per cycle.
    nframes = args.warmup + args.steps
    resident = min(nframes, MAX_RESIDENT)
    frames = [sd.synth(f) for f in range(resident)]
    out = torch.empty_like(frames[0][0])
    torch.cuda.synchronize()

    def step(f):
        c, nd, m = frames[f % resident]
        sd.denoise(c, nd, m, out)

    # Clock preconditioning, BEFORE the W warm-up steps and not part of them: a cold MI355X raises its shader clock over the first
    # ~50 ms of load, and the a-trous launches (VALU-bound: their time is cycles / clock) run 142 -> 119 us over the first 50
    # frames while T + V (HBM-bound) stays put (profiles/r04_clock_ramp.txt; round 3's driver run timed frames 5..25 of that
    # ramp).  `value` is the steady state the metric names, so the GPU is brought there first: PRE frames of the same loop, then
    # the history is dropped, so the W + K frames that follow are the sequence they always were (frame 0 = no history).
    pre_frames = int(os.environ.get("RMD_BENCH_PRECONDITION_FRAMES", "120"))
    # For the record, the SAME W + K frames as the process finds the GPU (what a run without the preconditioning reports: the first
    # W + K frames of the clock ramp); reported as `cold_start`, never as `value`.
    for f in range(args.warmup):
        step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_cold = time.perf_counter()
    for f in range(args.warmup, nframes):
        step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_cold = time.perf_counter() - t_cold
    if world > 1:
        tc = torch.tensor([t_cold], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        t_cold = float(tc.item())
    sd.reset_history()
    t_pre = time.perf_counter()
    for f in range(pre_frames):
        step(f)
    sd.reset_history()
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t_pre
    for f in range(args.warmup):
        step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The contract's region: EXACTLY K frames between barrier + synchronize on both sides, nothing else queued (an event
    # record between two launches costs ~6 us of idle GPU on this stack).
    t0 = time.perf_counter()
    for f in range(args.warmup, nframes):
        step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # SURVEY §8(d) also asks for the MEDIAN frame: a SECOND pass over the same K frames, every frame bracketed by HIP events
    # on its stream (serial frames only: the pipelined form spreads a frame over two streams); not part of `value`.
    frame_ms = None
    if not pipelined:
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
        for k, f in enumerate(range(args.warmup, nframes)):
            step(f)
            marks[k + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        frame_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_px = width * height * args.steps
    value = total_px / dt / 1e6
    result = {
        # BASELINE.json's metric string (the variance pass is part of "full SVGF": T + V + 5 x A)
        "metric": "Mpixels/s full SVGF (temporal+5 à-trous) at 1080p/4K; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        # per-frame HIP events on the frame's stream in a second pass over the same frames; the median is robust against the one
        # reprojection-miss frame per cycle through the resident sequence and against host hiccups, `value` is the wall-clock mean
        "ms_per_step_median": round(statistics.median(frame_ms), 4) if frame_ms else None,
        "mpix_s_at_median_frame": round(width * height / statistics.median(frame_ms) / 1e3, 1) if frame_ms else None,
        "cold_start": {"ms_per_step": round(t_cold / args.steps * 1e3, 4), "value": round(width * height * args.steps / t_cold / 1e6, 1),
                       "what": "the same W + K frames timed first, as the process finds the GPU (no preconditioning): the clock ramp of profiles/r04_clock_ramp.txt"},
        "preconditioning": {"frames": pre_frames, "ms": round(t_pre * 1e3, 1),
                            "what": "untimed frames of the same loop before the W warm-up steps (history reset afterwards): a cold GPU's "
                                    "shader clock takes ~50 ms of load to settle and the a-trous launches are clock-bound"},
        "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "frame": [width, height], "rows_per_gpu": rows_out,
                   "parallelism": f"row-strip x{world}", "passes": "T+V (one launch) + 5 x A (6 launches/frame)",
                   "frame_pipelining": "T+V of frame k+1 overlap A1..A4 of frame k (2 streams)" if pipelined else "none"},
        # 424 B/px is SURVEY §8(d)'s per-pass algorithmic count; with V's pass-through copy fused into T the
        # frame actually moves 360 B/px, which is the figure to hold against the HBM peak
        "effective_GBps_full_svgf": round(FULL_BYTES_PER_PX * total_px / dt / 1e9, 1),
        "moved_GBps_full_svgf": round(MOVED_BYTES_PER_PX * total_px / dt / 1e9, 1),
        "bytes_per_px": {"algorithmic_per_pass_sum": FULL_BYTES_PER_PX, "moved_with_V_fused_into_T": MOVED_BYTES_PER_PX},
    }
    if world > 1:
        result["halo_bytes_per_frame_rank0"] = {"history": sharding.halo_bytes(plan, width), "mid_frame": sharding.mid_halo_bytes(plan, width)}
        result["config"]["redundant_rows_per_side"] = {"inputs": plan.reach_in, "history": plan.reach_hist}
        result["config"]["exchange_iteration"] = plan.mid_iteration
        result["config"]["backend"] = backend
        result["config"]["exchanges"] = ("history halo after A0 + a-trous iteration %d's %d halo rows per side, both on a second stream "
                                         "beside the a-trous launches" % (plan.mid_iteration, plan.mid_rows)) if plan.mid_iteration >= 0 \
            else "history halo after A0 on a second stream; redundant rows instead of any exchange inside the frame"

    if rank == 0:
        result["roofline"] = measure_roofline(rmd, torch, sd.den, frames, width, rows_out, plan, args.roofline_reps,
                                              whole_frame_4k=(world == 1 and (width, height) == (3840, 2160)))
    if world == 1 and not args.no_other_sizes:
        # the other frame sizes north_star names, same pipeline, short runs (not the headline value)
        del frames, sd
        torch.cuda.empty_cache()
        # (1080p frames are 0.3 ms: more of them for a stable figure; 8K G-buffers are 1.3 GB per frame)
        def leg(name, fn):
            """The legs beside the contract's value: a failure in one of them (a pinned allocation, say) is reported in its
            place and does not take the line with it."""
            try:
                result[name] = fn()
            except Exception as e:                   # noqa: BLE001
                result[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

        leg("other_sizes", lambda: {f"{w}x{h}": other_size(rmd, torch, w, h, p, frames=n, warm=wu)
                                    for w, h, n, wu in ((1920, 1080, 48, 8), (7680, 4320, 16, 4))})
        leg("reference_api", lambda: reference_api_kernels(rmd, torch))
        leg("cornell_sequence_4k", lambda: cornell_sequence(rmd, torch, p))                              # BASELINE configs[4]
        leg("cornell_1080p", lambda: cornell_sequence(rmd, torch, p, width=1920, height=1080))          # BASELINE configs[1]: 1920x1080 Cornell, full SVGF
    if world > 1 and rank == 0 and not args.no_other_sizes:
        # the SAME 8K frame unsharded on rank 0's GPU: what the N-GPU figure is a speed-up over
        del frames, sd
        torch.cuda.empty_cache()
        one = other_size(rmd, torch, width, height, p, frames=8, warm=3)
        result["one_gpu_same_frame"] = dict(one, speedup=round(value / one["mpix_s"], 3))
        result["speedup_vs_one_gpu_same_frame"] = result["one_gpu_same_frame"]["speedup"]
        if backend != "nccl":
            result["speedup_note"] = "ranks share devices over gloo: a rehearsal of the code path; multi-GPU scaling is UNMEASURED ON HARDWARE"
    if world > 1:
        dist.barrier()
    if world > 1:
        dist.destroy_process_group()        # the other ranks are done: nobody waits in a collective while rank 0 times the CPU
    if rank == 0:
        # the scalar oracle on rank 0's host cores, for every N (the same bounded sample), after the timed region
        try:
            result["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline()
        except Exception as e:                       # noqa: BLE001  (the oracle failing to build must not take the GPU line with it)
            result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
