"""Host-side SVGF interface over the C ABI (rmd_svgf_* in include/rmd_api.h).

The reference only names SVGF (README.md:3-10); the pass semantics are SURVEY.md Appendix A.
torch supplies device memory (float32 CUDA tensors) and nothing else; every pass runs in the
HIP kernels of librmd.so.
"""
import ctypes as C
import sys

import torch

from ._lib import SvgfFrameDesc, SvgfParams, SynthDesc, check, lib
from .filter import _stream_ptr

# channels per pixel (0: a plane of shape [rows, width], one byte per pixel)
PLANE_CHANNELS = {"color": 4, "nd": 4, "motion": 2, "hist_color": 4, "hist_moments": 2, "hist_len": 0, "prev_nd": 4,
                  "t_color": 4, "t_moments": 2, "t_len": 0, "t_debug": 4, "v_color": 4, "hist_color_out": 4,
                  "out_color": 4}


def default_params() -> SvgfParams:
    p = SvgfParams()
    lib.rmd_svgf_default_params(C.byref(p))
    return p


def frame_reach(params: SvgfParams):
    """(input rows read, history rows read, hist_color rows produced, hist_moments rows produced)
    above/below the output rows of a whole-frame call (include/rmd_api.h rmd_svgf_frame_reach)."""
    r = (C.c_int * 4)()
    check(lib.rmd_svgf_frame_reach(C.byref(params), C.byref(r)))
    return tuple(int(v) for v in r)


def frame_mid_exchange(params: SvgfParams):
    """(exchange iteration or -1, rows per side that travel) of rmd_svgf_params.exchange_iteration."""
    r = (C.c_int * 2)()
    check(lib.rmd_svgf_frame_mid_exchange(C.byref(params), C.byref(r)))
    return int(r[0]), int(r[1])


def frame_iteration_reach(params: SvgfParams):
    """Rows above/below its strip on which a rank computes each a-trous iteration (rmd_svgf_frame_iteration_reach)."""
    r = (C.c_int * 8)()
    check(lib.rmd_svgf_frame_iteration_reach(C.byref(params), C.byref(r)))
    return [int(r[i]) for i in range(params.iterations)]


ATROUS_ALL, ATROUS_HEAD, ATROUS_INTERIOR, ATROUS_TAIL = 0, 1, 2, 3      # include/rmd_api.h RMD_ATROUS_*


def _ptr(t, name, rows, width, channels, dtype=torch.float32):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous CUDA {dtype} tensor")
    want = (rows, width, channels) if channels else (rows, width)
    if tuple(t.shape) != want:
        raise ValueError(f"{name}: shape {tuple(t.shape)} != {want}")
    return t.data_ptr()


def tile_flags_bytes(width, height):
    """RMD_TILE_FLAGS_BYTES of include/rmd_api.h."""
    return (((width + 63) // 64) * ((height + 3) // 4) + 3) // 4 * 4


def frame_desc(width, height, buf_row0=0, buf_rows=None, ping=(None, None), stats=None, tile_flags=None,
               **planes) -> SvgfFrameDesc:
    """Descriptor over torch planes of shape [buf_rows, width, C] (caller keeps them alive)."""
    buf_rows = height if buf_rows is None else buf_rows
    d = SvgfFrameDesc()
    d.width, d.height, d.buf_row0, d.buf_rows = width, height, buf_row0, buf_rows
    for name, t in planes.items():
        if name not in PLANE_CHANNELS:
            raise KeyError(name)
        dtype = torch.int32 if name == "t_debug" else torch.uint8 if PLANE_CHANNELS[name] == 0 else torch.float32
        setattr(d, name, _ptr(t, name, buf_rows, width, PLANE_CHANNELS[name], dtype))
    d.ping[0] = _ptr(ping[0], "ping[0]", buf_rows, width, 4)
    d.ping[1] = _ptr(ping[1], "ping[1]", buf_rows, width, 4)
    if stats is not None:
        if not (stats.is_cuda and stats.dtype == torch.float32 and stats.numel() >= 4):
            raise ValueError("stats: expected a CUDA float32 tensor with >= 4 elements")
        d.stats = stats.data_ptr()
    if tile_flags is not None:
        if not (tile_flags.is_cuda and tile_flags.dtype == torch.uint8 and tile_flags.numel() >= tile_flags_bytes(width, height)):
            raise ValueError("tile_flags: expected a CUDA uint8 tensor of tile_flags_bytes(width, height) elements")
        d.v_tile_flags = tile_flags.data_ptr()
    return d


def temporal(desc, params, row0, row1, stream=None):
    check(lib.rmd_svgf_temporal(C.byref(desc), C.byref(params), row0, row1, _stream_ptr(stream)))


def variance(desc, params, row0, row1, stream=None):
    check(lib.rmd_svgf_variance(C.byref(desc), C.byref(params), row0, row1, _stream_ptr(stream)))


def atrous(desc, params, iteration, src, dst, row0, row1, stream=None):
    rows, width = desc.buf_rows, desc.width
    check(lib.rmd_svgf_atrous(C.byref(desc), C.byref(params), iteration, _ptr(src, "in", rows, width, 4),
                              _ptr(dst, "out", rows, width, 4), row0, row1, _stream_ptr(stream)))


def frame(desc, params, row0, row1, stream=None):
    check(lib.rmd_svgf_frame(C.byref(desc), C.byref(params), row0, row1, _stream_ptr(stream)))


def synth_gbuffer(width, height, frame_index, buf_row0=0, buf_rows=None, seed=1234, pan=(1.25, -0.5),
                  device="cuda", want_albedo=False, out=None, stream=None):
    """Synthetic G-buffer of SURVEY §8(d) generated on the device: (color, nd, motion[, albedo])."""
    buf_rows = height if buf_rows is None else buf_rows
    if out is None:
        color = torch.empty((buf_rows, width, 4), dtype=torch.float32, device=device)
        nd = torch.empty_like(color)
        motion = torch.empty((buf_rows, width, 2), dtype=torch.float32, device=device)
    else:
        color, nd, motion = out
    albedo = torch.empty_like(color) if want_albedo else None
    d = SynthDesc(width, height, buf_row0, buf_rows, seed, frame_index, pan[0], pan[1])
    check(lib.rmd_synth_gbuffer(C.byref(d), color.data_ptr(), nd.data_ptr(), motion.data_ptr(),
                                albedo.data_ptr() if want_albedo else None, _stream_ptr(stream)))
    return (color, nd, motion, albedo) if want_albedo else (color, nd, motion)


def convert_u8_to_f32(src_u8, renormalize_xyz=False, w_value=-1.0, stream=None):
    """uint8 [H, W, 4] -> float32 [H, W, 4] (c/255, optional xyz renormalisation, w override)."""
    out = torch.empty(src_u8.shape, dtype=torch.float32, device=src_u8.device)
    check(lib.rmd_convert_u8_to_f32(src_u8.data_ptr(), out.data_ptr(), src_u8.shape[0] * src_u8.shape[1],
                                    int(renormalize_xyz), float(w_value), _stream_ptr(stream)))
    return out


def convert_f32_to_u8(src_f32, albedo=None, stream=None):
    out = torch.empty(src_f32.shape, dtype=torch.uint8, device=src_f32.device)
    check(lib.rmd_convert_f32_to_u8(src_f32.data_ptr(), None if albedo is None else albedo.data_ptr(),
                                    out.data_ptr(), src_f32.shape[0] * src_f32.shape[1], _stream_ptr(stream)))
    return out


def demodulate(radiance, albedo, eps=1e-3, out=None, stream=None):
    """illumination = radiance / max(albedo, eps) per rgb channel, w passed through (rmd_demodulate)."""
    out = torch.empty_like(radiance) if out is None else out
    check(lib.rmd_demodulate(radiance.data_ptr(), albedo.data_ptr(), out.data_ptr(), radiance.shape[0] * radiance.shape[1],
                             float(eps), _stream_ptr(stream)))
    return out


class SvgfDenoiser:
    """Cross-frame SVGF state for one device / one row strip.

    Same plane routing as the C context (rmd_svgf_context_*), but the planes are torch tensors
    so a row-strip deployment can hand their halo rows to torch.distributed (RCCL).  Planes hold
    global rows [buf_row0, buf_row0+buf_rows); `denoise` produces rows [row0,row1).

    pipelined=True software-pipelines consecutive frames over two HIP streams: T+V of frame k+1
    (HBM-bound) is issued on a second stream as soon as frame k's history is complete (after its
    a-trous iteration `hist_iteration`) and runs underneath frame k's remaining a-trous iterations
    (ALU-bound).  Same kernels, same bits; results are valid after `synchronize()`.
    """

    def __init__(self, width, height, buf_row0=0, buf_rows=None, params=None, device="cuda", debug=False,
                 collect_stats=False, pipelined=False):
        self.width, self.height = width, height
        self.buf_row0 = buf_row0
        self.buf_rows = height if buf_rows is None else buf_rows
        self.params = params if params is not None else default_params()
        self.device = device

        def plane():
            return torch.zeros((self.buf_rows, width, 4), dtype=torch.float32, device=device)

        self.hist_color = [plane(), plane()]
        # the luminance moments (m1, m2) as float2 and the history length as one byte per pixel: 9 of the 16 bytes a float4
        # (m1, m2, h, 0) would move per tap and per pixel written (include/rmd_api.h rmd_svgf_frame_desc)
        self.hist_moments = [torch.zeros((self.buf_rows, width, 2), dtype=torch.float32, device=device) for _ in range(2)]
        self.hist_len = [torch.zeros((self.buf_rows, width), dtype=torch.uint8, device=device) for _ in range(2)]
        self.t_color, self.v_color = plane(), plane()
        self.ping = [plane(), plane()]
        self.t_debug = torch.zeros((self.buf_rows, width, 4), dtype=torch.int32, device=device) if debug else None
        # frame statistics accumulated by the V pass (diagnostics; costs atomics, so opt-in)
        self.stats = torch.zeros(4, dtype=torch.float32, device=device) if collect_stats else None
        self.tile_flags = torch.zeros(tile_flags_bytes(width, height), dtype=torch.uint8, device=device)
        self.cur = 0
        self.has_history = False
        self.prev_nd = None
        self.pipelined = pipelined
        self._ev_hist = C.c_void_p()
        check(lib.rmd_event_create(C.byref(self._ev_hist)))
        if pipelined:
            # The a-trous stream is the high-priority queue (measured: no effect on how the dispatcher
            # arbitrates between T's pending workgroups and a new a-trous launch; kept as the intent).
            self.stream_a = torch.cuda.Stream(device=device, priority=-1)       # a-trous iterations
            self.stream_b = torch.cuda.Stream(device=device, priority=0)        # T + V (+ the history halo exchange)
            self._ev_tv = C.c_void_p()
            check(lib.rmd_event_create(C.byref(self._ev_tv)))
            self._hist_recorded = False

    def __del__(self):
        if sys.is_finalizing():            # the HIP runtime may already be gone: leave streams and events to the process exit
            return
        for ev in (getattr(self, "_ev_hist", None), getattr(self, "_ev_tv", None)):
            if ev:
                lib.rmd_event_destroy(ev)

    def reset_history(self):
        self.has_history = False
        self.prev_nd = None

    def history(self):
        """(hist_color, hist_moments, hist_len) the NEXT denoise call reads."""
        return self.hist_color[self.cur], self.hist_moments[self.cur], self.hist_len[self.cur]

    def synchronize(self):
        if self.pipelined:
            self.stream_a.synchronize()
            self.stream_b.synchronize()
        else:
            torch.cuda.current_stream().synchronize()

    def describe(self, color, nd, motion, out, ahead=False):
        """The frame descriptor of the next denoise call -- or, with ahead=<the next call's nd>, of the call AFTER it (its
        history planes are the ones the next call writes, its prev_nd the next call's nd)."""
        cur = self.cur ^ 1 if ahead is not False else self.cur
        prev_nd = ahead if ahead is not False else self.prev_nd
        use_hist = ahead is not False or (self.has_history and self.prev_nd is not None)
        return frame_desc(
            self.width, self.height, self.buf_row0, self.buf_rows,
            color=color, nd=nd, motion=motion,
            hist_color=self.hist_color[cur] if use_hist else None,
            hist_moments=self.hist_moments[cur] if use_hist else None,
            hist_len=self.hist_len[cur] if use_hist else None,
            prev_nd=prev_nd if use_hist else None,
            t_color=self.t_color, t_moments=self.hist_moments[cur ^ 1], t_len=self.hist_len[cur ^ 1], t_debug=self.t_debug,
            v_color=self.v_color, hist_color_out=self.hist_color[cur ^ 1],
            ping=(self.ping[0], self.ping[1]), out_color=out, stats=self.stats, tile_flags=self.tile_flags)

    def iteration_plane(self, iteration, out=None):
        """The plane a-trous iteration `iteration` of the NEXT denoise call writes (rmd_svgf_frame's routing)."""
        n, pp = self.params.iterations, 0
        for k in range(iteration + 1):
            if k == n - 1:
                dst = out
            elif k == self.params.hist_iteration:
                dst = self.hist_color[self.cur ^ 1]
            else:
                dst, pp = self.ping[pp], pp ^ 1
        return dst

    def denoise(self, color, nd, motion, out=None, row0=None, row1=None, stream=None, before_tv=None, hooks=None, _tv_done=False):
        """One frame.  `nd` is borrowed until the next call (it becomes prev_nd).  `before_tv` is
        called just before T is launched, on the stream T runs on (a row-strip deployment completes
        its history halo there).  Serial form: runs on `stream` (default: torch's current stream).
        Pipelined form: runs on two side streams; the inputs must stay untouched until synchronize().

        `hooks` (serial form; a row-strip deployment, sharding.ShardedDenoiser) may have
          hist_ready(event)  called on the host once the launches that complete next frame's history planes are queued;
                             `event` (a hipEvent_t) is recorded behind them on the frame's stream
          mid_ready(plane)   params.exchange_iteration = X >= 0: called after iterations 0..X-1 and the BOUNDARY rows of X are
                             queued, with X's output plane; the hook starts the neighbour exchange of those rows on another
                             stream (X's interior rows are queued next and run meanwhile)
          mid_wait()         called before the launches that read that halo (makes the frame's stream wait for it)."""
        if out is None:
            out = torch.empty_like(color)
        row0 = max(self.buf_row0, 0) if row0 is None else row0
        row1 = min(self.buf_row0 + self.buf_rows, self.height) if row1 is None else row1
        d = self.describe(color, nd, motion, out)
        tv_done = _tv_done          # (experiments.NextFrameDenoiser: T + V of this frame already ran inside the previous frame's launches)
        is_strip = row0 > 0 or row1 < self.height
        if not self.pipelined:
            if before_tv is not None:
                before_tv()
            # the caller's CURRENT torch stream unless told otherwise (NULL would be unordered with a
            # `with torch.cuda.stream(s)` block around this call)
            s_ptr = _stream_ptr(torch.cuda.current_stream() if stream is None else stream)
            mid = frame_mid_exchange(self.params)[0]
            p = self.params
            if mid >= 0 and is_strip and (hooks is None or not hasattr(hooks, "mid_ready") or not hasattr(hooks, "mid_wait")):
                # without the exchange the iteration behind `mid` would read halo rows of the previous frame
                raise ValueError(f"rows [{row0},{row1}) are a strip and exchange_iteration = {mid}: denoise() needs hooks.mid_ready / "
                                 "hooks.mid_wait to exchange the halo rows (sharding.ShardedDenoiser)")
            # the history-ready event only where somebody waits for it (a row-strip deployment): an event record between two
            # launches costs ~6 us of idle GPU on this stack (tools/frame_gaps.py)
            ev_hist = self._ev_hist if hooks is not None and hasattr(hooks, "hist_ready") else None
            if not tv_done:
                check(lib.rmd_svgf_frame_tv(C.byref(d), C.byref(p), row0, row1, s_ptr))
            if mid < 0:
                check(lib.rmd_svgf_frame_atrous(C.byref(d), C.byref(p), row0, row1, s_ptr, ev_hist))
                if hooks is not None and hasattr(hooks, "hist_ready"):
                    hooks.hist_ready(ev_hist)
            else:
                hist_in_head = p.hist_iteration < mid           # (the exchanged iteration itself completes with its INTERIOR part)
                check(lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), row0, row1, s_ptr, ev_hist, ATROUS_HEAD))
                if hooks is not None and hasattr(hooks, "mid_ready"):
                    hooks.mid_ready(self.iteration_plane(mid, out))
                if hist_in_head and hooks is not None and hasattr(hooks, "hist_ready"):
                    hooks.hist_ready(ev_hist)
                check(lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), row0, row1, s_ptr, ev_hist, ATROUS_INTERIOR))
                if p.hist_iteration == mid and hooks is not None and hasattr(hooks, "hist_ready"):
                    hooks.hist_ready(ev_hist)
                if hooks is not None and hasattr(hooks, "mid_wait"):
                    hooks.mid_wait()
                check(lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), row0, row1, s_ptr, ev_hist, ATROUS_TAIL))
                if p.hist_iteration > mid and hooks is not None and hasattr(hooks, "hist_ready"):
                    hooks.hist_ready(ev_hist)
        else:
            if tv_done:
                raise ValueError("the next-frame side job and the pipelined form do not mix")
            if frame_mid_exchange(self.params)[0] >= 0:
                raise ValueError("exchange_iteration >= 0 needs the serial (single-stream) frame: pipelined=False")
            sa, sb = self.stream_a, self.stream_b
            # The side streams keep reading color / nd / motion and writing `out` after this call returns (nd
            # until the NEXT frame's T has run): tell torch's caching allocator, so that a caller who drops or
            # recycles these tensors does not get their memory handed out again while kernels are in flight.
            # Results are valid, and inputs may be overwritten, after synchronize().
            for t in (color, nd, motion, out):
                t.record_stream(sa)
                t.record_stream(sb)
            sb.wait_stream(torch.cuda.current_stream())            # the caller's inputs
            if self._hist_recorded:                                # frame k's history complete (after its A_hist)
                check(lib.rmd_stream_wait_event(sb.cuda_stream, self._ev_hist))
            with torch.cuda.stream(sb):
                if before_tv is not None:
                    before_tv()
                check(lib.rmd_svgf_frame_tv(C.byref(d), C.byref(self.params), row0, row1, sb.cuda_stream))
            check(lib.rmd_event_record(self._ev_tv, sb.cuda_stream))
            check(lib.rmd_stream_wait_event(sa.cuda_stream, self._ev_tv))
            check(lib.rmd_svgf_frame_atrous(C.byref(d), C.byref(self.params), row0, row1, sa.cuda_stream, self._ev_hist))
            self._hist_recorded = True
        self.cur ^= 1
        self.has_history = True
        self.prev_nd = nd
        return out


class GBufferDenoiser:
    """SVGF on the reference's own frame descriptor (include/gbuffer.h:6-14): one call per frame, uint8 [H, W, 4] render /
    albedo / normal planes in, uint8 denoised out (rmd_svgf_gbuffer_frame; the 8-bit ends run inside the frame's first and
    last launch).  The cross-frame state lives in the C context (rmd_svgf_context_*), not in torch tensors."""

    def __init__(self, width, height, params=None, albedo_eps=1.0 / 255.0, debug=False, device="cuda"):
        self.width, self.height = width, height
        self.params = params if params is not None else default_params()
        self.albedo_eps = float(albedo_eps)
        self._ctx = C.c_void_p()
        check(lib.rmd_svgf_context_create(width, height, 0, height, C.byref(self._ctx)))
        self.t_debug = torch.zeros((height, width, 4), dtype=torch.int32, device=device) if debug else None
        if debug:
            check(lib.rmd_svgf_context_set_debug_plane(self._ctx, self.t_debug.data_ptr()))

    def __del__(self):
        if sys.is_finalizing():            # the HIP runtime may already be gone
            return
        ctx = getattr(self, "_ctx", None)
        if ctx:
            lib.rmd_svgf_context_destroy(ctx)
            self._ctx = None

    def reset_history(self, stream=None):
        check(lib.rmd_svgf_context_reset_history(self._ctx, _stream_ptr(stream)))

    def frame(self, render, albedo, normal, denoised=None, motion=None, stream=None):
        """One frame; returns `denoised` (uint8 [H, W, 4]).  motion: float32 [H, W, 2] (current -> previous) or None = static."""
        from .filter import make_gbuffer
        if denoised is None:
            denoised = torch.empty_like(render)
        if tuple(render.shape) != (self.height, self.width, 4):
            raise ValueError(f"render: shape {tuple(render.shape)} != {(self.height, self.width, 4)}")
        g = make_gbuffer(render, denoised, normal=normal, albedo=albedo)
        m = _ptr(motion, "motion", self.height, self.width, 2)
        s_ptr = _stream_ptr(torch.cuda.current_stream() if stream is None else stream)
        check(lib.rmd_svgf_gbuffer_frame(g, self._ctx, C.byref(self.params), m, self.albedo_eps, s_ptr))
        return denoised

    def history(self):
        """Copies of (hist_color, hist_moments, hist_len) the NEXT frame reads: float32 [H, W, 4], float32 [H, W, 2], uint8 [H, W]."""
        hc, hm, hl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.rmd_svgf_context_history(self._ctx, C.byref(hc), C.byref(hm), C.byref(hl)))
        out = []
        torch.cuda.current_stream().synchronize()
        for ptr, shape, dtype in ((hc, (self.height, self.width, 4), torch.float32), (hm, (self.height, self.width, 2), torch.float32),
                                  (hl, (self.height, self.width), torch.uint8)):
            t = torch.empty(shape, dtype=dtype, device="cuda")
            check(lib.rmd_memcpy_d2d(t.data_ptr(), ptr, t.numel() * t.element_size(), None))
            out.append(t)
        check(lib.rmd_device_sync())
        return tuple(out)
