"""The C-ABI row-strip planner (rmd_strip_plan_make / rmd_halo_plan / rmd_halo_bytes, csrc/strips.hip) against
the Python one the torch.distributed path uses (raymarchdenoisercuda_amd/sharding.py): same strips, same
buffers, same halo steps in the same order.  Pure arithmetic, no GPU.  The RCCL exchange itself is
exercised on the GPU box (loop-back on one device)."""
import ctypes as C

import pytest
import torch

from raymarchdenoisercuda_amd import sharding
from raymarchdenoisercuda_amd._lib import HaloStep, StripPlan, lib


def c_plan(rmd, height, world, rank, p):
    plan = StripPlan()
    rmd.check(lib.rmd_strip_plan_make(height, world, rank, C.byref(p), C.byref(plan)))
    return plan


def c_steps(rmd, plan, fn=None):
    steps = (HaloStep * 12)()                                       # RMD_HALO_MAX_STEPS
    n = C.c_int()
    rmd.check((fn or lib.rmd_halo_plan)(C.byref(plan), steps, 12, C.byref(n)))
    kind = {HaloStep.RECV: "recv", HaloStep.SEND: "send"}
    name = {0: "color", 1: "moments", 2: "mid", 3: "len"}
    return [(kind[s.kind], name[s.plane], s.row_lo, s.row_hi, s.peer) for s in steps[:n.value]]


@pytest.mark.parametrize("exchange_iteration", [-1, 3, 2, 0])
@pytest.mark.parametrize("height,world", [(4320, 8), (4320, 4), (4320, 2), (2160, 4), (1000, 3), (420, 2), (300, 1), (4321, 8)])
def test_c_planner_equals_python_planner(rmd, height, world, exchange_iteration):
    p = rmd.default_params()
    p.max_motion_rows = 8
    p.exchange_iteration = exchange_iteration
    reach, mid = rmd.svgf.frame_reach(p), rmd.svgf.frame_mid_exchange(p)
    assert mid == ((exchange_iteration, {3: 32, 2: 48, 0: 60}[exchange_iteration]) if exchange_iteration >= 0 else (-1, 0))
    for rank in range(world):
        want = sharding.make_plan(height, world, rank, reach, mid)
        got = c_plan(rmd, height, world, rank, p)
        for f in ("height", "world", "rank", "row0", "row1", "buf_row0", "buf_rows", "reach_in", "reach_hist", "have_color", "have_moments",
                  "mid_iteration", "mid_rows"):
            assert getattr(got, f) == getattr(want, f), (rank, f)
        assert c_steps(rmd, got) == sharding.halo_plan(want), rank
        assert c_steps(rmd, got, lib.rmd_mid_halo_plan) == sharding.mid_halo_plan(want), rank
        assert lib.rmd_halo_bytes(C.byref(got), 7680) == sharding.halo_bytes(want, 7680)
        r0, r1 = C.c_int(), C.c_int()
        rmd.check(lib.rmd_strip_rows(height, world, rank, C.byref(r0), C.byref(r1)))
        assert (r0.value, r1.value) == sharding.strip_rows(height, world, rank)


def test_other_parameters_change_the_plan_the_same_way(rmd):
    p = rmd.default_params()
    p.max_motion_rows, p.iterations, p.hist_iteration = 2, 3, 2
    reach = rmd.svgf.frame_reach(p)
    for rank in range(4):
        want = sharding.make_plan(2160, 4, rank, reach)
        got = c_plan(rmd, 2160, 4, rank, p)
        assert c_steps(rmd, got) == sharding.halo_plan(want)
        assert (got.buf_row0, got.buf_rows) == (want.buf_row0, want.buf_rows)


def test_reach_with_one_exchange_inside_the_frame(rmd):
    """exchange_iteration = 3 (SURVEY §8e "Halo sizes"): A3 on the strip's own rows, its 32-row halo travels; T / V / A0..A2
    run on 32 fewer rows per side than with redundant rows only."""
    p = rmd.default_params()
    p.max_motion_rows = 8
    assert rmd.svgf.frame_reach(p) == (66, 73, 60, 65)
    p.exchange_iteration = 3
    assert rmd.svgf.frame_reach(p) == (34, 41, 28, 33) and rmd.svgf.frame_mid_exchange(p) == (3, 32)
    p.exchange_iteration = 4                                                             # the last iteration: nothing to exchange for
    with pytest.raises(rmd.RmdError):
        rmd.svgf.frame_reach(p)
    p.exchange_iteration = 0
    p.hist_iteration = 0                                                                 # the history iteration IS the exchanged one:
    assert rmd.svgf.frame_reach(p)[2] == 60                                              # its own rows + the 60 received


def test_short_strips_and_bad_arguments_are_rejected(rmd):
    p = rmd.default_params()
    plan = StripPlan()
    assert lib.rmd_strip_plan_make(400, 8, 0, C.byref(p), C.byref(plan)) == -5          # RMD_E_ROWS, like sharding.make_plan
    with pytest.raises(ValueError):
        sharding.make_plan(400, 8, 0, rmd.svgf.frame_reach(p))
    assert lib.rmd_strip_plan_make(4320, 8, 8, C.byref(p), C.byref(plan)) == -3          # rank outside the world
    assert lib.rmd_strip_plan_make(4320, 8, 0, None, C.byref(plan)) == -1
    n = C.c_int()
    rmd.check(lib.rmd_strip_plan_make(4320, 8, 3, C.byref(p), C.byref(plan)))
    rmd.check(lib.rmd_halo_plan(C.byref(plan), None, 0, C.byref(n)))                      # counting form
    assert n.value == 12                                                                  # 3 planes x (recv + send) x 2 neighbours
    steps = (HaloStep * 2)()
    assert lib.rmd_halo_plan(C.byref(plan), steps, 2, C.byref(n)) == -4 and n.value == 12  # RMD_E_BUFFER, count still reported


def test_exchange_without_neighbours_needs_no_communicator(rmd):
    p = rmd.default_params()
    plan = c_plan(rmd, 300, 1, 0, p)
    rmd.check(lib.rmd_halo_exchange(None, C.byref(plan), 64, None, None, None, None))


@pytest.mark.gpu
def test_rccl_loopback_exchange_on_one_gpu(rmd, cuda):
    """ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd through the C ABI with a one-rank communicator:
    rows [10,14) of hist_color (float4) are sent to self into rows [20,24), rows [3,5) of hist_moments (float2) into [30,32), and
    the same rows of hist_len (uint8: travels as bytes)."""
    assert lib.rmd_comm_available() == 1, "librccl.so not found on the GPU box"
    comm = C.c_void_p()
    rmd.check(lib.rmd_comm_create_all(1, None, C.byref(comm)))
    width, rows = 96, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    hc = torch.rand((rows, width, 4), device="cuda", generator=g)
    hm = torch.rand((rows, width, 2), device="cuda", generator=g)
    hl = torch.randint(0, 256, (rows, width), dtype=torch.uint8, device="cuda", generator=g)
    want_c, want_m, want_l = hc.clone(), hm.clone(), hl.clone()
    want_c[20:24] = hc[10:14]
    want_m[30:32] = hm[3:5]
    want_l[30:32] = hl[3:5]
    steps = (HaloStep * 6)(HaloStep(HaloStep.RECV, 0, 120, 124, 0), HaloStep(HaloStep.SEND, 0, 110, 114, 0),
                           HaloStep(HaloStep.RECV, 1, 130, 132, 0), HaloStep(HaloStep.SEND, 1, 103, 105, 0),
                           HaloStep(HaloStep.RECV, HaloStep.PLANE_HIST_LEN, 130, 132, 0), HaloStep(HaloStep.SEND, HaloStep.PLANE_HIST_LEN, 103, 105, 0))
    stream = torch.cuda.current_stream().cuda_stream
    rmd.check(lib.rmd_halo_exchange_steps(comm, 0, steps, 6, 100, rows, width, hc.data_ptr(), hm.data_ptr(), hl.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.equal(hc, want_c) and torch.equal(hm, want_m) and torch.equal(hl, want_l)
    bad = (HaloStep * 1)(HaloStep(HaloStep.SEND, 0, 90, 95, 0))
    assert lib.rmd_halo_exchange_steps(comm, 0, bad, 1, 100, rows, width, hc.data_ptr(), hm.data_ptr(), hl.data_ptr(), stream) == -5
    # the mid-frame plane through the same group: rows [0,4) of an a-trous plane to self into rows [36,40)
    mid = torch.rand((rows, width, 4), device="cuda", generator=g)
    want_mid = mid.clone()
    want_mid[36:40] = mid[0:4]
    planes = (C.c_void_p * 4)(None, None, mid.data_ptr(), None)
    msteps = (HaloStep * 2)(HaloStep(HaloStep.RECV, HaloStep.PLANE_MID, 136, 140, 0), HaloStep(HaloStep.SEND, HaloStep.PLANE_MID, 100, 104, 0))
    rmd.check(lib.rmd_exchange_steps(comm, 0, msteps, 2, 100, rows, width, planes, stream))
    torch.cuda.synchronize()
    assert torch.equal(mid, want_mid)
    assert lib.rmd_exchange_steps(comm, 0, (HaloStep * 1)(HaloStep(HaloStep.SEND, 5, 100, 104, 0)), 1, 100, rows, width, planes, stream) == -3
    rmd.check(lib.rmd_comm_destroy(comm))
