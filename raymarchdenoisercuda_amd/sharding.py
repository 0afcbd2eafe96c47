"""Row-strip sharding of one frame across the GPUs of a node.

Not in the reference (single device, no communication: SURVEY §0.4).  One process per GPU,
torch.distributed with backend "nccl" (= RCCL over xGMI).  A frame of H rows is cut into
contiguous full-width row strips; each rank runs the whole SVGF pipeline on its strip plus the
rows later passes tap (redundant rows, no per-pass exchange), so within a frame there is no
communication at all.  The only cross-rank data is the temporal feedback: next frame's history
halo.  A rank already holds hist_color on +-reach[2] rows and hist_moments on +-reach[3] rows of
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
    have_moments: int  # hist_moments rows this rank produces itself beyond the strip


def strip_rows(height, world, rank):
    base, rem = divmod(height, world)
    row0 = rank * base + min(rank, rem)
    return row0, row0 + base + (1 if rank < rem else 0)


def make_plan(height, world, rank, reach) -> StripPlan:
    """reach = svgf.frame_reach(params) = (reach_in, reach_hist, have_color, have_moments)."""
    reach_in, reach_hist, have_color, have_moments = reach
    row0, row1 = strip_rows(height, world, rank)
    if world > 1:
        smallest = height // world
        if smallest < reach_hist:
            raise ValueError(f"strips of {smallest} rows are shorter than the history reach {reach_hist}: "
                             "use fewer ranks, a taller frame or a smaller max_motion_rows")
    reach = max(reach_in, reach_hist)
    b0, b1 = max(0, row0 - reach), min(height, row1 + reach)
    return StripPlan(height, world, rank, row0, row1, b0, b1 - b0, reach_in, reach_hist, have_color, have_moments)


def _rows(plane, plan, lo, hi):
    """View of global rows [lo,hi) of a strip plane (contiguous: full-width rows)."""
    return plane[lo - plan.buf_row0: hi - plan.buf_row0]


def halo_plan(plan: StripPlan):
    """The per-frame exchange of one rank as plain data: a list of
    (kind, plane, lo, hi, peer) with kind in {"recv", "send"}, plane in {"color", "moments"} and
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
    for name, have in (("color", plan.have_color), ("moments", plan.have_moments)):
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


def halo_ops(plan: StripPlan, hist_color, hist_moments, group=None):
    """torch.distributed P2P ops for halo_plan(plan) on this rank's history planes."""
    planes = {"color": hist_color, "moments": hist_moments}
    ops = []
    for kind, name, lo, hi, peer in halo_plan(plan):
        fn = dist.irecv if kind == "recv" else dist.isend
        ops.append(dist.P2POp(fn, _rows(planes[name], plan, lo, hi), peer, group))
    return ops


def exchange_history_halo(plan: StripPlan, hist_color, hist_moments, group=None):
    """Blocking form: post the batched isend/irecv group and wait for it.

    With RCCL ("nccl") the device rows travel directly over xGMI.  With gloo and device planes (the
    one-GPU rehearsal of the multi-rank path) the rows are staged through host memory.
    """
    steps = halo_plan(plan)
    if not steps:
        return 0
    planes = {"color": hist_color, "moments": hist_moments}
    staged = hist_color.is_cuda and dist.get_backend(group) == "gloo"
    if not staged:
        for req in dist.batch_isend_irecv(halo_ops(plan, hist_color, hist_moments, group)):
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


def halo_bytes(plan: StripPlan, width):
    """Bytes this rank receives per frame (for reporting)."""
    total = 0
    for have in (plan.have_color, plan.have_moments):
        rows = max(0, plan.reach_hist - have)
        if plan.rank > 0:
            total += min(rows, plan.row0) * width * 16
        if plan.rank < plan.world - 1:
            total += min(rows, plan.height - plan.row1) * width * 16
    return total


class ShardedDenoiser:
    """Per-rank driver: strip plan + SvgfDenoiser + the per-frame history halo exchange.

    The halo of frame k's history is completed lazily, right before frame k+1's temporal pass and
    on the stream that pass runs on; with pipelined=True that is the second stream, so the exchange
    (like T and V) runs underneath frame k's remaining a-trous iterations.
    """

    def __init__(self, width, height, params=None, device="cuda", group=None, rank=None, world=None, pipelined=False):
        from . import svgf  # needs librmd.so; the plan/exchange helpers above do not
        self.svgf = svgf
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.params = params if params is not None else svgf.default_params()
        self.plan = make_plan(height, self.world, self.rank, svgf.frame_reach(self.params))
        self.width, self.height = width, height
        self.den = svgf.SvgfDenoiser(width, height, self.plan.buf_row0, self.plan.buf_rows, self.params, device,
                                     pipelined=pipelined)
        self.exchange = self.world > 1 and dist.is_initialized()
        self._halo_pending = False

    def synth(self, frame_index, out=None, **kw):
        return self.svgf.synth_gbuffer(self.width, self.height, frame_index, self.plan.buf_row0, self.plan.buf_rows,
                                       out=out, **kw)

    def _complete_halo(self):
        if self._halo_pending:
            hc, hm = self.den.history()
            exchange_history_halo(self.plan, hc, hm, self.group)
            self._halo_pending = False

    def denoise(self, color, nd, motion, out=None):
        """Strip rows [row0,row1) of `out` (valid after synchronize())."""
        out = self.den.denoise(color, nd, motion, out, self.plan.row0, self.plan.row1,
                               before_tv=self._complete_halo if self.exchange else None)
        self._halo_pending = self.exchange
        return out

    def synchronize(self):
        self.den.synchronize()
