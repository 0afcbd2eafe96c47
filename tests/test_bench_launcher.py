"""bench.py --gpus N without a launcher: the parent starts the N ranks itself (before anything touches
the GPU), relays rank 0's JSON line and reports a failing rank through its exit status.  Rehearsed on
CPU with --rehearse-launcher (ranks rendezvous over gloo and reduce one number; no GPU work)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)                      # the shape of the driver's N = 1 command: no launcher environment
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=600)


def test_parent_starts_the_ranks_and_relays_rank0():
    r = run_bench("--gpus", "3", "--rehearse-launcher")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got == {"launcher_rehearsal": True, "n_gpus": 3, "max_over_ranks": 3.0}


def test_a_failing_rank_fails_the_run():
    r = run_bench("--gpus", "2", "--rehearse-launcher", RMD_BENCH_FAIL_RANK="1")
    assert r.returncode == 3
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launcher_does_not_import_torch_or_the_library_in_the_parent():
    """The parent must not touch the GPU: everything before launch_ranks() is standard library."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    for line in head.splitlines():
        assert not line.startswith(("import torch", "from torch", "import raymarchdenoisercuda_amd")), line
    body = src[src.index("def main()"):]
    assert body.index("launch_ranks(args)") < body.index("import torch")



@pytest.mark.gpu
def test_two_rank_strong_scaling_bench_runs_without_a_launcher():
    """`python bench.py --gpus 2` exactly as the driver calls N = 1 (no launcher, no WORLD_SIZE): the parent starts
    both ranks, they cut the fixed 8K frame of BASELINE configs[3] into two strips and exchange the history
    halo every frame.  On the one-GPU box the two ranks share the device, so the exchange runs over gloo
    (RCCL refuses two ranks on one GPU); with one GPU per rank the same code path uses backend "nccl"."""
    r = run_bench("--gpus", "2", "--steps", "3", "--warmup", "2", "--no-other-sizes", "--roofline-reps", "3",
                  RMD_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert d["config"]["frame"] == [7680, 4320] and d["config"]["rows_per_gpu"] == 2160
    assert "configs[3]" in d["config"]["workload"]
    # default for N > 1: one exchange inside the frame (a-trous iteration 3's 32 halo rows); history rows (28, 41] of
    # hist_color (float4) and (33, 41] of hist_moments (float2) and hist_len (uint8)
    assert d["config"]["exchange_iteration"] == 3
    assert d["halo_bytes_per_frame_rank0"] == {"history": 13 * 7680 * 16 + 8 * 7680 * (8 + 1), "mid_frame": 32 * 7680 * 16}
    assert d["cpu_baseline"]["value"] > 0 and d["ms_per_step_median"] > 0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
