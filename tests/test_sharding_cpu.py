"""Row-strip sharding logic on CPU: plan arithmetic and the torch.distributed halo exchange
over gloo with world_size 2 and 3 (the N>1 path of bench.py; RCCL is used on GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raymarchdenoisercuda_amd import sharding

REACH = (66, 73, 60, 65)          # frame_reach(default params with max_motion_rows=8)


def test_strips_partition_the_frame():
    for h, w in ((4320, 8), (2160, 4), (1000, 3), (420, 2)):
        rows = [sharding.strip_rows(h, w, r) for r in range(w)]
        assert rows[0][0] == 0 and rows[-1][1] == h
        assert all(rows[i][1] == rows[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1


def test_plan_buffers_cover_every_read():
    for world in (1, 2, 4, 8):
        for rank in range(world):
            p = sharding.make_plan(4320, world, rank, REACH)
            assert p.buf_row0 == max(0, p.row0 - 73) and p.buf_row0 + p.buf_rows == min(4320, p.row1 + 73)
            assert p.row1 - p.row0 == 4320 // world


def test_short_strips_are_rejected():
    with pytest.raises(ValueError):
        sharding.make_plan(400, 8, 0, REACH)


def test_sends_and_receives_pair_up():
    """Every recv of a rank is matched by the same rows being sent by the owner, in order."""
    world, height = 4, 2160
    plans = [sharding.make_plan(height, world, r, REACH) for r in range(world)]
    steps = [sharding.halo_plan(p) for p in plans]
    for r in range(world):
        for peer in (r - 1, r + 1):
            if not 0 <= peer < world:
                continue
            recvs = [(n, lo, hi) for k, n, lo, hi, q in steps[r] if k == "recv" and q == peer]
            sends = [(n, lo, hi) for k, n, lo, hi, q in steps[peer] if k == "send" and q == r]
            assert recvs == sends and recvs
            for _, lo, hi in recvs:
                assert plans[peer].row0 <= lo and hi <= plans[peer].row1      # sent rows are owned rows
    assert sharding.halo_plan(sharding.make_plan(height, 1, 0, REACH)) == []
    assert sharding.halo_bytes(plans[1], 3840) == 2 * (13 * 16 + 8 * (8 + 1)) * 3840          # hist_color 16 B/px, hist_moments 8 + hist_len 1


REACH_X3 = (34, 41, 28, 33)       # the same with exchange_iteration = 3: T / V / A0..A2 on 32 fewer rows per side
MID_X3 = (3, 32)                  # frame_mid_exchange: iteration 3's output, 32 rows per side


def test_mid_frame_exchange_plan_pairs_up():
    """exchange_iteration = 3: every rank receives the 32 rows beyond either end of its strip from the neighbour that
    OWNS them (it computed iteration 3 on its own rows only), in matching order; buffers cover the received rows."""
    world, height = 8, 4320
    plans = [sharding.make_plan(height, world, r, REACH_X3, MID_X3) for r in range(world)]
    steps = [sharding.mid_halo_plan(p) for p in plans]
    for r in range(world):
        p = plans[r]
        assert p.buf_row0 == max(0, p.row0 - 41) and p.buf_row0 + p.buf_rows == min(height, p.row1 + 41)
        for peer in (r - 1, r + 1):
            if not 0 <= peer < world:
                continue
            recvs = [(lo, hi) for k, n, lo, hi, q in steps[r] if k == "recv" and q == peer]
            sends = [(lo, hi) for k, n, lo, hi, q in steps[peer] if k == "send" and q == r]
            assert recvs == sends and len(recvs) == 1 and recvs[0][1] - recvs[0][0] == 32
            lo, hi = recvs[0]
            assert plans[peer].row0 <= lo and hi <= plans[peer].row1          # sent rows are the sender's own rows
            assert hi == p.row0 or lo == p.row1                               # and touch this rank's strip
            assert p.buf_row0 <= lo and hi <= p.buf_row0 + p.buf_rows
    assert sharding.mid_halo_plan(sharding.make_plan(height, 1, 0, REACH_X3, MID_X3)) == []
    assert sharding.mid_halo_plan(sharding.make_plan(height, 8, 3, REACH)) == []          # no exchange_iteration: nothing travels
    assert sharding.mid_halo_bytes(plans[1], 7680) == 2 * 32 * 7680 * 16
    assert sharding.mid_halo_bytes(plans[0], 7680) == 32 * 7680 * 16
    assert sharding.halo_bytes(plans[1], 7680) == 2 * (13 * 16 + 8 * (8 + 1)) * 7680     # history halo: rows (28, 41] of hist_color, (33, 41] of hist_moments + hist_len
    with pytest.raises(ValueError):
        sharding.make_plan(240, 8, 0, (34, 20, 28, 33), MID_X3)                            # 30-row strips < 32 exchanged rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


PLANE_CHANNELS = {0: 4, 1: 2, 2: 4, 3: 0}          # hist_color float4, hist_moments float2, the mid plane float4, hist_len uint8 [rows, width]


def _truth(rows, width, plane_id):
    """What global row y of a plane must contain."""
    if PLANE_CHANNELS[plane_id] == 0:
        y = torch.arange(rows[0], rows[1], dtype=torch.int64).view(-1, 1)
        x = torch.arange(width, dtype=torch.int64).view(1, -1)
        return ((y * 7 + x * 3 + 1) % 251).to(torch.uint8)
    y = torch.arange(rows[0], rows[1], dtype=torch.float32).view(-1, 1, 1)
    x = torch.arange(width, dtype=torch.float32).view(1, -1, 1)
    ch = torch.arange(PLANE_CHANNELS[plane_id], dtype=torch.float32).view(1, 1, -1)
    return y * 1000.0 + x + ch * 0.125 + plane_id * 0.5


def _worker(rank, world, port, height, width, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = sharding.make_plan(height, world, rank, REACH)
        planes = {}
        for pid, have in ((0, plan.have_color), (1, plan.have_moments), (3, plan.have_moments)):
            t = torch.full((plan.buf_rows, width, PLANE_CHANNELS[pid]), -1.0) if PLANE_CHANNELS[pid] else torch.full((plan.buf_rows, width), 255, dtype=torch.uint8)
            lo, hi = max(0, plan.row0 - have), min(height, plan.row1 + have)   # what the rank computed itself
            t[lo - plan.buf_row0:hi - plan.buf_row0] = _truth((lo, hi), width, pid)
            planes[pid] = t
        for _ in range(2):                                                   # twice: the exchange is per frame
            n = sharding.exchange_history_halo(plan, planes[0], planes[1], planes[3])
        ok = n == len(sharding.halo_plan(plan))
        lo, hi = max(0, plan.row0 - plan.reach_hist), min(height, plan.row1 + plan.reach_hist)
        for pid, t in planes.items():
            ok = ok and torch.equal(t[lo - plan.buf_row0:hi - plan.buf_row0], _truth((lo, hi), width, pid))
        # the mid-frame exchange: a plane that holds only the rank's OWN rows gets the 32 rows on either side
        plan = sharding.make_plan(height, world, rank, REACH_X3, MID_X3)
        t = torch.full((plan.buf_rows, width, 4), -1.0)
        t[plan.row0 - plan.buf_row0:plan.row1 - plan.buf_row0] = _truth((plan.row0, plan.row1), width, 2)
        n = sharding.exchange_mid_halo(plan, t)
        ok = ok and n == len(sharding.mid_halo_plan(plan))
        lo, hi = max(0, plan.row0 - 32), min(height, plan.row1 + 32)
        ok = ok and torch.equal(t[lo - plan.buf_row0:hi - plan.buf_row0], _truth((lo, hi), width, 2))
        result[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_over_gloo(world):
    height, width = 600, 48
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [result.get(r) for r in range(world)] == [True] * world
