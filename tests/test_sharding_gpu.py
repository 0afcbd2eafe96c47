"""Row-strip deployment with REAL processes and torch.distributed on the GPU box: 2 and 3 ranks
share cuda:0 over gloo (RCCL refuses two ranks on one device; on the 8-GPU node the same code runs
one rank per GPU with backend "nccl").  Every rank checks its strip bit-for-bit against the
single-device result it computes itself (SURVEY §8e "Parity across G")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, width, height, frames, exchange_iteration, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raymarchdenoisercuda_amd as rmd
        from raymarchdenoisercuda_amd import sharding
        torch.cuda.set_device(0)
        p = rmd.default_params()
        p.max_motion_rows = 8
        single = rmd.SvgfDenoiser(width, height, params=rmd.SvgfParams.from_buffer_copy(p))
        p.exchange_iteration = exchange_iteration          # the strips: redundant rows only, or one exchange inside the frame
        sd = sharding.ShardedDenoiser(width, height, params=p, rank=rank, world=world)
        ok = True
        for f in range(frames):
            c, nd, m = rmd.svgf.synth_gbuffer(width, height, f)
            want = single.denoise(c, nd, m)
            cs, nds, ms = sd.synth(f)
            got = sd.denoise(cs, nds, ms)
            torch.cuda.synchronize()
            a, b = sd.plan.row0, sd.plan.row1
            ok = ok and torch.equal(got[a - sd.plan.buf_row0:b - sd.plan.buf_row0], want[a:b])
        result[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange_iteration", [-1, 3])
@pytest.mark.parametrize("world", [2, 3])
def test_strips_over_torch_distributed_match_single_device(cuda, world, exchange_iteration):
    ctx = mp.get_context("spawn")
    result = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 160, 420, 4, exchange_iteration, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert [result.get(r) for r in range(world)] == [True] * world
