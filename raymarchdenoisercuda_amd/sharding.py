"""Row-strip sharding of one frame across the GPUs of a node.

Not in the reference (single device, no communication: SURVEY §0.4).  One process per GPU,
torch.distributed with backend "nccl" (= RCCL over xGMI).  A frame of H rows is cut into
contiguous full-width row strips; each rank runs the whole SVGF pipeline on its strip plus the
rows later passes tap (redundant rows, no per-pass exchange), so within a frame there is no
communication at all.  The only cross-rank data is the temporal feedback: next frame's history
halo.  A rank already holds hist_color on +-reach[2] rows and hist_moments / hist_len on +-reach[3] rows of
its strip, bit-identical to its neighbours' copies, so only the rows beyond that, up to
reach[1], travel: point-to-point isend/irecv with rank+-1, one batched group per frame (one
direct xGMI link per neighbour; no all-reduce anywhere).

The exchange code only touches torch tensors and torch.distributed, so it is covered on CPU with
the gloo backend (tests/test_sharding_cpu.py).
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class StripPlan:
    height: int
    world: int
    rank: int
    row0: int          # owned output rows [row0, row1)
    row1: int
    buf_row0: int      # rows held by every plane of this rank [buf_row0, buf_row0 + buf_rows)
    buf_rows: int
    reach_in: int      # current-frame input rows read beyond the strip
    reach_hist: int    # history rows read beyond the strip
    have_color: int    # hist_color rows this rank produces itself beyond the strip
    have_moments: int  # hist_moments / hist_len rows this rank produces itself beyond the strip
    mid_iteration: int = -1   # a-trous iteration whose output crosses ranks INSIDE a frame (rmd_svgf_params.exchange_iteration)
    mid_rows: int = 0         # rows per side that travel in that exchange


def strip_rows(height, world, rank):
    base, rem = divmod(height, world)
    row0 = rank * base + min(rank, rem)
    return row0, row0 + base + (1 if rank < rem else 0)


def make_plan(height, world, rank, reach, mid=(-1, 0)) -> StripPlan:
    """reach = svgf.frame_reach(params) = (reach_in, reach_hist, have_color, have_moments);
    mid = svgf.frame_mid_exchange(params) = (exchange iteration or -1, rows per side)."""
    reach_in, reach_hist, have_color, have_moments = reach
    mid_iteration, mid_rows = mid
    row0, row1 = strip_rows(height, world, rank)
    if world > 1:
        smallest = height // world
        if smallest < reach_hist:
            raise ValueError(f"strips of {smallest} rows are shorter than the history reach {reach_hist}: "
                             "use fewer ranks, a taller frame or a smaller max_motion_rows")
        if smallest < mid_rows:
            raise ValueError(f"strips of {smallest} rows are shorter than the {mid_rows} rows of the mid-frame exchange: "
                             "use fewer ranks or a later exchange_iteration")
    reach = max(reach_in, reach_hist, mid_rows)
    b0, b1 = max(0, row0 - reach), min(height, row1 + reach)
    return StripPlan(height, world, rank, row0, row1, b0, b1 - b0, reach_in, reach_hist, have_color, have_moments,
                     mid_iteration, mid_rows)


def _rows(plane, plan, lo, hi):
    """View of global rows [lo,hi) of a strip plane (contiguous: full-width rows)."""
    return plane[lo - plan.buf_row0: hi - plan.buf_row0]


def halo_plan(plan: StripPlan):
    """The per-frame exchange of one rank as plain data: a list of
    (kind, plane, lo, hi, peer) with kind in {"recv", "send"}, plane in {"color", "moments", "len"} (hist_color float4,
    hist_moments float2, hist_len uint8: T writes the last two on the same rows) and
    [lo,hi) GLOBAL rows.  From rank-1 a rank needs rows [row0-reach_hist, row0-have); from rank+1
    rows [row1+have, row1+reach_hist); the sends mirror the neighbours' needs.  Rows outside the
    frame do not exist and are skipped.  recv/send pairs of two neighbours appear in matching
    order, so posting them as one batched group cannot deadlock.
    """
    steps = []
    if plan.world == 1:
        return steps
    up, down = plan.rank - 1, plan.rank + 1
    H = plan.height
    for name, have in (("color", plan.have_color), ("moments", plan.have_moments), ("len", plan.have_moments)):
        if have >= plan.reach_hist:
            continue
        if up >= 0:
            lo, hi = max(0, plan.row0 - plan.reach_hist), max(0, plan.row0 - have)
            if hi > lo:
                steps.append(("recv", name, lo, hi, up))
            # rank-1's lower need [row0+have, row0+reach_hist) lies in this rank's strip
            lo, hi = min(H, plan.row0 + have), min(H, plan.row0 + plan.reach_hist)
            if hi > lo:
                steps.append(("send", name, lo, hi, up))
        if down < plan.world:
            lo, hi = min(H, plan.row1 + have), min(H, plan.row1 + plan.reach_hist)
            if hi > lo:
                steps.append(("recv", name, lo, hi, down))
            lo, hi = max(0, plan.row1 - plan.reach_hist), max(0, plan.row1 - have)
            if hi > lo:
                steps.append(("send", name, lo, hi, down))
    return steps


def mid_halo_plan(plan: StripPlan):
    """The exchange INSIDE a frame (rmd_svgf_params.exchange_iteration = X): iteration X is computed on the strip's own
    rows only; the mid_rows rows of its output beyond either end of the strip come from the neighbour that computed
    them as ITS own rows.  Same step format as halo_plan, plane "mid"."""
    steps = []
    if plan.world == 1 or plan.mid_iteration < 0 or plan.mid_rows <= 0:
        return steps
    up, down, R, H = plan.rank - 1, plan.rank + 1, plan.mid_rows, plan.height
    if up >= 0:
        steps.append(("recv", "mid", max(0, plan.row0 - R), plan.row0, up))
        steps.append(("send", "mid", plan.row0, min(H, plan.row0 + R), up))
    if down < plan.world:
        steps.append(("recv", "mid", plan.row1, min(H, plan.row1 + R), down))
        steps.append(("send", "mid", max(0, plan.row1 - R), plan.row1, down))
    return [s for s in steps if s[3] > s[2]]


def halo_ops(plan: StripPlan, hist_color, hist_moments, hist_len, group=None):
    """torch.distributed P2P ops for halo_plan(plan) on this rank's history planes."""
    planes = {"color": hist_color, "moments": hist_moments, "len": hist_len}
    ops = []
    for kind, name, lo, hi, peer in halo_plan(plan):
        fn = dist.irecv if kind == "recv" else dist.isend
        ops.append(dist.P2POp(fn, _rows(planes[name], plan, lo, hi), peer, group))
    return ops


def exchange_history_halo(plan: StripPlan, hist_color, hist_moments, hist_len, group=None):
    """Blocking form: post the batched isend/irecv group and wait for it.

    With RCCL ("nccl") the device rows travel directly over xGMI.  With gloo and device planes (the
    one-GPU rehearsal of the multi-rank path) the rows are staged through host memory.
    """
    return _exchange(plan, halo_plan(plan), {"color": hist_color, "moments": hist_moments, "len": hist_len}, group)


def exchange_mid_halo(plan: StripPlan, mid_plane, group=None):
    """The mid-frame exchange of mid_halo_plan(plan) on the output plane of iteration plan.mid_iteration."""
    return _exchange(plan, mid_halo_plan(plan), {"mid": mid_plane}, group)


def _exchange(plan, steps, planes, group):
    if not steps:
        return 0
    first = next(iter(planes.values()))
    staged = first.is_cuda and dist.get_backend(group) == "gloo"
    if not staged:
        ops = [dist.P2POp(dist.irecv if kind == "recv" else dist.isend, _rows(planes[name], plan, lo, hi), peer, group)
               for kind, name, lo, hi, peer in steps]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return len(steps)
    ops, landing = [], []
    for kind, name, lo, hi, peer in steps:
        rows = _rows(planes[name], plan, lo, hi)
        if kind == "send":
            ops.append(dist.P2POp(dist.isend, rows.cpu(), peer, group))
        else:
            host = torch.empty(rows.shape, dtype=rows.dtype)
            landing.append((rows, host))
            ops.append(dist.P2POp(dist.irecv, host, peer, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for rows, host in landing:
        rows.copy_(host)
    return len(steps)


PLANE_PIXEL_BYTES = {"color": 16, "moments": 8, "len": 1, "mid": 16}


def mid_halo_bytes(plan: StripPlan, width):
    """Bytes this rank receives in the mid-frame exchange."""
    return sum((hi - lo) * width * 16 for kind, _, lo, hi, _ in mid_halo_plan(plan) if kind == "recv")


def halo_bytes(plan: StripPlan, width):
    """History bytes this rank receives per frame (for reporting)."""
    return sum((hi - lo) * width * PLANE_PIXEL_BYTES[name] for kind, name, lo, hi, _ in halo_plan(plan) if kind == "recv")


class _NoExchange:
    """Exchange hooks that exchange nothing (ShardedDenoiser(timing_only_no_exchange=True))."""

    def mid_ready(self, plane):
        pass

    def mid_wait(self):
        pass


class ShardedDenoiser:
    """Per-rank driver: strip plan + SvgfDenoiser + the neighbour exchanges of a frame.

    Serial frames (the default): the 7 launches of a frame run on one stream; the exchanges run on a second one,
    ordered by events, underneath the a-trous iterations --
      * next frame's HISTORY halo right after the history iteration (A_0) has been queued, awaited in front of the
        next frame's temporal pass;
      * with params.exchange_iteration = X >= 0, the MID-FRAME halo of iteration X's output: X runs on its boundary rows
        first (what the neighbours wait for), they travel while X's interior rows are computed, and X+1 starts on the
        whole strip once the halo is in.
    pipelined=True is the older two-stream form (T+V of frame k+1 beside the a-trous iterations of frame k; the history
    halo completed lazily on the T+V stream); it does not support exchange_iteration.
    """

    def __init__(self, width, height, params=None, device="cuda", group=None, rank=None, world=None, pipelined=False,
                 timing_only_no_exchange=False):
        from . import svgf  # needs librmd.so; the plan/exchange helpers above do not
        self.svgf = svgf
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.params = params if params is not None else svgf.default_params()
        self.plan = make_plan(height, self.world, self.rank, svgf.frame_reach(self.params), svgf.frame_mid_exchange(self.params))
        self.width, self.height = width, height
        self.den = svgf.SvgfDenoiser(width, height, self.plan.buf_row0, self.plan.buf_rows, self.params, device,
                                     pipelined=pipelined)
        self.exchange = self.world > 1 and dist.is_initialized()
        # tools/strip_probe.py: what ONE rank's launches cost without a process group -- the halo rows are never delivered, the
        # outputs are not a frame; SvgfDenoiser.denoise refuses that unless it is told so through (empty) exchange hooks
        self.timing_only = bool(timing_only_no_exchange) and not self.exchange
        self.pipelined = pipelined
        self._halo_pending = False
        self._hist_done = self._mid_done = None
        if self.exchange and not pipelined and torch.device(device).type == "cuda":
            self.comm_stream = torch.cuda.Stream(device=device)
            self._hist_done, self._mid_done = torch.cuda.Event(), torch.cuda.Event()
        else:
            self.comm_stream = None           # CPU planes (gloo tests): the exchanges run inline

    def synth(self, frame_index, out=None, **kw):
        return self.svgf.synth_gbuffer(self.width, self.height, frame_index, self.plan.buf_row0, self.plan.buf_rows,
                                       out=out, **kw)

    # ---- pipelined form: the history halo lazily, on the stream T runs on
    def _complete_halo(self):
        if self._halo_pending:
            exchange_history_halo(self.plan, *self.den.history(), self.group)
            self._halo_pending = False

    # ---- serial form: hooks of SvgfDenoiser.denoise
    def before_tv(self):
        if self._halo_pending and self._hist_done is not None:
            torch.cuda.current_stream().wait_event(self._hist_done)
        self._halo_pending = False

    def _on_comm_stream(self, after_event, fn, done):
        from ._lib import check, lib
        check(lib.rmd_stream_wait_event(self.comm_stream.cuda_stream, after_event))
        with torch.cuda.stream(self.comm_stream):
            fn()
            done.record(self.comm_stream)

    def hist_ready(self, event):
        """Next frame's history planes are complete behind `event`: exchange their halo rows now, beside the remaining
        a-trous iterations."""
        den = self.den
        hc, hm, hl = den.hist_color[den.cur ^ 1], den.hist_moments[den.cur ^ 1], den.hist_len[den.cur ^ 1]      # what the NEXT frame reads
        if self.comm_stream is None:
            exchange_history_halo(self.plan, hc, hm, hl, self.group)
        else:
            self._on_comm_stream(event, lambda: exchange_history_halo(self.plan, hc, hm, hl, self.group), self._hist_done)
        self._halo_pending = True

    def mid_ready(self, plane):
        if self.comm_stream is None:
            exchange_mid_halo(self.plan, plane, self.group)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            exchange_mid_halo(self.plan, plane, self.group)
            self._mid_done.record(self.comm_stream)

    def mid_wait(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_event(self._mid_done)

    def denoise(self, color, nd, motion, out=None):
        """Strip rows [row0,row1) of `out` (valid after synchronize())."""
        if not self.exchange:
            hooks = _NoExchange() if self.timing_only and self.plan.mid_iteration >= 0 else None
            return self.den.denoise(color, nd, motion, out, self.plan.row0, self.plan.row1, hooks=hooks)
        if self.pipelined:
            out = self.den.denoise(color, nd, motion, out, self.plan.row0, self.plan.row1, before_tv=self._complete_halo)
            self._halo_pending = True
            return out
        return self.den.denoise(color, nd, motion, out, self.plan.row0, self.plan.row1, before_tv=self.before_tv, hooks=self)

    def synchronize(self):
        self.den.synchronize()
        if self.comm_stream is not None:
            self.comm_stream.synchronize()

    def reset_history(self):
        """The next frame is a first frame again (every rank must call it at the same point of the sequence)."""
        self.synchronize()
        self.den.reset_history()
        self._halo_pending = False
