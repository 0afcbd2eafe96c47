#!/usr/bin/env python3
"""How many host threads does the scalar oracle actually get on this box?  (bench.py cpu_baseline)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

print("hardware threads", orc.hardware_threads(), "affinity", len(os.sched_getaffinity(0)), "os.cpu_count", os.cpu_count())
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(path, open(path).read().strip())
    except OSError as e:
        print(path, "-", e.__class__.__name__)
w, h = 1920, 1080
p = orc.default_params()
c, nd, m = orc.synth_gbuffer(w, h, 0)
f0 = orc.Frame(w, h, c, nd, m, None, None, None, debug=False)
orc.frame(f0, p, threads=16)
c1, nd1, m1 = orc.synth_gbuffer(w, h, 1)
for threads in (8, 16, 32, 64, 128, 256):
    f1 = orc.Frame(w, h, c1, nd1, m1, *f0.history(), debug=False)
    t0 = time.perf_counter()
    orc.frame(f1, p, threads=threads)
    dt = time.perf_counter() - t0
    print(f"threads {threads:4d}: {dt:.3f} s  {w * h / dt / 1e6:.2f} Mpix/s")
