"""Frame pipelining (two HIP streams) must not change a single bit: the same kernels run on the same
data, only T+V of frame k+1 is issued underneath the a-trous iterations of frame k."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def run_sequence(rmd, width, height, frames, pipelined, p):
    den = rmd.SvgfDenoiser(width, height, params=p, pipelined=pipelined)
    inputs = [rmd.svgf.synth_gbuffer(width, height, f) for f in range(frames)]
    torch.cuda.synchronize()
    outs = [torch.empty_like(inputs[0][0]) for _ in range(frames)]
    for f in range(frames):                                   # no host sync between frames
        c, nd, m = inputs[f]
        den.denoise(c, nd, m, outs[f])
    den.synchronize()
    torch.cuda.synchronize()
    return outs, den


@pytest.mark.parametrize("width,height", [(300, 200), (1920, 1080)])
def test_pipelined_frames_equal_serial_frames(rmd, cuda, width, height):
    p = rmd.default_params()
    p.max_motion_rows = 8
    serial, den_s = run_sequence(rmd, width, height, 8, False, p)
    piped, den_p = run_sequence(rmd, width, height, 8, True, p)
    for f, (a, b) in enumerate(zip(serial, piped)):
        assert torch.equal(a, b), f"frame {f}: {(a != b).sum().item()} values differ"
    for a, b in zip(den_s.history(), den_p.history()):
        assert torch.equal(a, b)


@pytest.mark.experiments
@pytest.mark.parametrize("workgroups", [1, 7, 256, 5000])
def test_persistent_tv_grids_equal_one_workgroup_per_tile(rmd, cuda, workgroups):
    """rmd_svgf_params.tv_workgroups: T and V as N persistent workgroups walking the tiles give the
    bits of the one-workgroup-per-tile launch, for whole frames and for a row range (strip)."""
    width, height = 333, 150
    p0 = rmd.default_params()
    p0.max_motion_rows = 8
    p1 = rmd.SvgfParams.from_buffer_copy(p0)
    p1.tv_workgroups = workgroups
    for rows in (None, (37, 101)):
        dens = [rmd.SvgfDenoiser(width, height, params=p, debug=True) for p in (p0, p1)]
        for f in range(4):
            c, nd, m = rmd.svgf.synth_gbuffer(width, height, f)
            outs = []
            for den in dens:
                out = torch.zeros_like(c)
                if rows is None:
                    den.denoise(c, nd, m, out)
                else:
                    den.denoise(c, nd, m, out, row0=rows[0], row1=rows[1])
                outs.append(out)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[1]), f"frame {f} rows {rows}"
            assert torch.equal(dens[0].t_debug, dens[1].t_debug)
            assert torch.equal(dens[0].tile_flags, dens[1].tile_flags)
            for a, b in zip(dens[0].history(), dens[1].history()):
                assert torch.equal(a, b)


def test_pipelined_repeatable(rmd, cuda):
    """Race check: many frames, several runs, always the same bits."""
    p = rmd.default_params()
    ref, _ = run_sequence(rmd, 640, 360, 12, True, p)
    for _ in range(3):
        again, _ = run_sequence(rmd, 640, 360, 12, True, p)
        assert all(torch.equal(a, b) for a, b in zip(ref, again))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, width, height, frames, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raymarchdenoisercuda_amd as rmd
        from raymarchdenoisercuda_amd import sharding
        torch.cuda.set_device(0)
        p = rmd.default_params()
        p.max_motion_rows = 8
        sd = sharding.ShardedDenoiser(width, height, params=p, rank=rank, world=world, pipelined=True)
        single = rmd.SvgfDenoiser(width, height, params=p)
        strips = [sd.synth(f) for f in range(frames)]
        outs = [torch.empty_like(strips[0][0]) for _ in range(frames)]
        torch.cuda.synchronize()
        for f in range(frames):
            sd.denoise(*strips[f], outs[f])
        sd.synchronize()
        ok = True
        a, b = sd.plan.row0, sd.plan.row1
        for f in range(frames):
            c, nd, m = rmd.svgf.synth_gbuffer(width, height, f)
            want = single.denoise(c, nd, m)
            torch.cuda.synchronize()
            ok = ok and torch.equal(outs[f][a - sd.plan.buf_row0:b - sd.plan.buf_row0], want[a:b])
        result[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_pipelined_row_strips_over_torch_distributed(cuda):
    """Two ranks (sharing cuda:0 over gloo), pipelined, halo exchange on the T stream: strips still
    equal the single-device frames bit for bit."""
    world = 2
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 160, 420, 5, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert [result.get(r) for r in range(world)] == [True] * world


@pytest.mark.parametrize("stats", [False, True])
def test_scratch_planes_poisoned_with_nan_change_nothing(rmd, cuda, stats):
    """t_color is scratch inside rmd_svgf_frame (with statistics: the separate T and V launches write all of it; without:
    T and V run as one launch and never touch it), v_color and the ping planes are written before they are read: poisoning
    all of them with NaN before every frame must leave the output and the history finite and bit-equal to an unpoisoned run."""
    width, height, frames = 333, 150, 4
    p = rmd.default_params()
    p.max_motion_rows = 8
    dens = [rmd.SvgfDenoiser(width, height, params=p, collect_stats=stats) for _ in range(2)]
    for f in range(frames):
        c, nd, m = rmd.svgf.synth_gbuffer(width, height, f)
        for t in (dens[1].t_color, dens[1].v_color, dens[1].ping[0], dens[1].ping[1]):
            t.fill_(float("nan"))
        outs = [den.denoise(c, nd, m, torch.full_like(c, float("nan"))) for den in dens]
        torch.cuda.synchronize()
        assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1]), f"frame {f}"
        for a, b in zip(dens[0].history(), dens[1].history()):
            assert torch.isfinite(b).all() and torch.equal(a, b), f"frame {f} history"


def test_frame_parts_are_refused_without_an_exchange_iteration(rmd, cuda):
    import ctypes as C
    width, height = 128, 64
    p = rmd.default_params()
    den = rmd.SvgfDenoiser(width, height, params=p)
    c, nd, m = rmd.svgf.synth_gbuffer(width, height, 0)
    d = den.describe(c, nd, m, torch.empty_like(c))
    for part in (rmd.svgf.ATROUS_HEAD, rmd.svgf.ATROUS_INTERIOR, rmd.svgf.ATROUS_TAIL):
        assert rmd.lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), 0, height, None, None, part) == -3      # RMD_E_PARAM
    assert rmd.lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), 0, height, None, None, 7) == -3
    p.exchange_iteration = 4                                   # the last iteration has nothing behind it
    assert rmd.lib.rmd_svgf_frame_atrous_part(C.byref(d), C.byref(p), 0, height, None, None, rmd.svgf.ATROUS_ALL) == -3
    p.exchange_iteration = 3                                   # a whole frame needs no halo: the parts in order ARE the frame
    den2 = rmd.SvgfDenoiser(width, height, params=p)
    want = rmd.SvgfDenoiser(width, height, params=rmd.default_params()).denoise(c, nd, m)
    got = den2.denoise(c, nd, m)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("width,height", [(300, 200), (1920, 1080)])
def test_frames_replayed_from_a_graph_equal_eager_frames(rmd, cuda, width, height):
    """rmd_graph_*: two frames (the history planes ping-pong) captured on a side stream and replayed give bit for bit what
    the same frame sequence gives eagerly."""
    p = rmd.default_params()
    p.max_motion_rows = 8
    a, b = rmd.svgf.synth_gbuffer(width, height, 0), rmd.svgf.synth_gbuffer(width, height, 1)

    eager = rmd.SvgfDenoiser(width, height, params=p)
    want = torch.empty_like(a[0])
    for _ in range(4):                                   # A B | A B A B A B
        eager.denoise(*a, out=want)
        eager.denoise(*b, out=want)
    torch.cuda.synchronize()

    den = rmd.SvgfDenoiser(width, height, params=p)
    got = torch.empty_like(a[0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        den.denoise(*a, out=got)                         # eager: first use on this stream, history planes valid
        den.denoise(*b, out=got)
        side.synchronize()
    with rmd.capture(side) as g:
        den.denoise(*a, out=got)
        den.denoise(*b, out=got)
    side.synchronize()
    assert not torch.equal(got, want)                    # the captured frames have not run
    for _ in range(3):
        g.launch()
    side.synchronize()
    assert torch.equal(got, want), f"{(got != want).sum().item()} values differ"
    for x, y in zip(eager.history(), den.history()):
        assert torch.equal(x, y)
    g.destroy()
    g.destroy()                                          # idempotent


def test_graph_argument_errors(rmd, cuda):
    import ctypes as C
    with pytest.raises(ValueError):
        rmd.capture(torch.cuda.default_stream())
    assert rmd.lib.rmd_graph_capture_begin(None) == -1                     # RMD_E_NULL: the legacy default stream
    assert rmd.lib.rmd_graph_launch(None, None) == -1
    assert rmd.lib.rmd_graph_destroy(None) == 0
    h = C.c_void_p()
    side = torch.cuda.Stream()
    assert rmd.lib.rmd_graph_capture_end(C.c_void_p(side.cuda_stream), C.byref(h)) != 0        # no capture in progress
    assert not h


@pytest.mark.experiments
@pytest.mark.parametrize("width,height", [(300, 200), (1920, 1080)])
def test_next_frames_temporal_pass_inside_the_atrous_launches(rmd, cuda, width, height):
    """rmd_svgf_frame_atrous_next (experiments build; it lost on time): the temporal pass of frame k+1
    as a side job of frame k's a-trous launches gives the serial frames bit for bit, whoever ends up running a tile."""
    p = rmd.default_params()
    p.max_motion_rows = 8
    frames = 6
    serial, den_s = run_sequence(rmd, width, height, frames, False, p)
    inputs = [rmd.svgf.synth_gbuffer(width, height, f) for f in range(frames)]
    from raymarchdenoisercuda_amd.experiments import NextFrameDenoiser
    den = NextFrameDenoiser(width, height, params=p)
    outs = [torch.empty_like(inputs[0][0]) for _ in range(frames)]
    for f in range(frames):
        den.denoise(*inputs[f], out=outs[f], next_frame=inputs[f + 1] if f + 1 < frames else None)
    torch.cuda.synchronize()
    for f, (a, b) in enumerate(zip(serial, outs)):
        assert torch.equal(a, b), f"frame {f}: {(a != b).sum().item()} values differ"
    for a, b in zip(den_s.history(), den.history()):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):                       # the frame announced as next_frame must be the one that follows
        den.denoise(*inputs[0], out=outs[0], next_frame=inputs[1])
        den.denoise(*inputs[2], out=outs[2])


def test_next_frame_side_job_is_refused_by_the_product_build(rmd, cuda):
    if rmd.HAS_EXPERIMENTS:
        pytest.skip("experiments build")
    p = rmd.default_params()
    from raymarchdenoisercuda_amd.experiments import NextFrameDenoiser
    den = NextFrameDenoiser(128, 64, params=p)
    a, b = rmd.svgf.synth_gbuffer(128, 64, 0), rmd.svgf.synth_gbuffer(128, 64, 1)
    with pytest.raises(rmd.RmdError) as e:
        den.denoise(*a, next_frame=b)
    assert e.value.code == -6
