"""The work decomposition of the default a-trous kernel (rmd_debug_atrous_plan: host arithmetic, no GPU):
every output row of every strip and lattice is produced by exactly one workgroup, the workgroup count
fits the resident slots the planner aimed at, and the mixed band counts (some strips one band more than
the others) are only used where the two sizes stay close."""
import ctypes as C

import numpy as np
import pytest

CW, ADV, SLOTS_PER_CU = 128, 4, 3          # StreamCfg<S, 2>: strip width, lattice rows per step, workgroups per CU


def plan(rmd, width, height, row0, row1, iteration, cus=256):
    out = (C.c_int * 8)()
    assert rmd.lib.rmd_debug_atrous_plan(width, height, row0, row1, iteration, cus, out) == 0, rmd.last_error()
    keys = ("nblocks", "nstrips", "band_base", "band_h", "band_h_hi", "n_hi", "nblocks_hi", "per_xcd")
    return dict(zip(keys, out))


def coverage(p, step, width, row0, row1):
    """Replays the decode of atrous_stream_kernel for every workgroup; returns how often each
    (row, strip) cell is produced (lattices are interleaved rows, so row index covers them)."""
    hits = np.zeros((row1, p["nstrips"]), np.int32)
    for L in range(p["nblocks"]):
        r = L % step
        if L < p["nblocks_hi"]:
            t = L // step
            e = t % p["n_hi"]
            strip = e if e < p["n_hi"] // 2 else p["nstrips"] - (p["n_hi"] - e)
            band, bh = t // p["n_hi"], p["band_h_hi"]
        else:
            t = (L - p["nblocks_hi"]) // step
            n_lo = p["nstrips"] - p["n_hi"]
            strip, band, bh = p["n_hi"] // 2 + t % n_lo, t // n_lo, p["band_h"]
        yb = p["band_base"] + band * bh
        lo, hi = max(yb, row0), min(yb + bh, row1)
        ybase = yb + r
        jlo = (lo - ybase + step - 1) // step if lo > ybase else 0
        jhi = (hi - ybase + step - 1) // step if hi > ybase else 0
        for j in range(jlo, jhi):
            hits[ybase + j * step, strip] += 1
    return hits


@pytest.mark.parametrize("width,height,row0,row1", [
    (3840, 2160, 0, 2160), (1920, 1080, 0, 1080), (523, 301, 0, 301), (64, 48, 0, 48), (300, 70, 17, 53),
    (7680, 4320, 2160 - 60, 2700 + 60),           # an interior rank's first a-trous launch of an 8-strip 8K run
    (7680, 4320, 0, 4320),
])
def test_every_row_is_produced_exactly_once(rmd, width, height, row0, row1):
    for it in range(5):
        step = 1 << it
        p = plan(rmd, width, height, row0, row1, it)
        assert p["nstrips"] == (width + CW - 1) // CW
        assert p["band_base"] % (2 * step) == 0 and p["band_base"] <= row0
        assert p["band_h"] % (step * ADV) == 0 and p["band_h_hi"] % (step * ADV) == 0
        assert 0 <= p["n_hi"] <= p["nstrips"] and p["per_xcd"] * 8 >= p["nblocks"]
        hits = coverage(p, step, width, row0, row1)
        assert (hits[row0:row1] == 1).all(), f"iteration {it}: rows produced {np.unique(hits[row0:row1])} times"
        assert (hits[:row0] == 0).all()


def test_4k_launches_fill_the_resident_slots(rmd):
    """3840x2160 on 256 CUs: steps 1, 2, 4 use the mixed band counts to reach 768 workgroups (one round);
    steps 8 and 16 would have to mix 3 with 4 (or 1 with 2) bands and keep one size."""
    slots = SLOTS_PER_CU * 256
    for it, want_mixed in ((0, True), (1, True), (2, True), (3, False), (4, False)):
        p = plan(rmd, 3840, 2160, 0, 2160, it)
        assert (p["n_hi"] > 0) == want_mixed, (it, p)
        if want_mixed:
            assert p["nblocks"] == slots, (it, p)
            assert p["band_h_hi"] < p["band_h"] and p["band_h_hi"] * 100 >= p["band_h"] * 85
        else:
            assert p["band_h_hi"] == p["band_h"] or p["n_hi"] == 0


def test_fewer_cus_mean_fewer_slots(rmd):
    """rmd_svgf_params.atrous_cus (a CU-partition stream): the plan targets that many CUs."""
    full = plan(rmd, 3840, 2160, 0, 2160, 1, cus=256)
    part = plan(rmd, 3840, 2160, 0, 2160, 1, cus=224)
    assert full["nblocks"] <= 3 * 256 and part["nblocks"] <= 2 * 3 * 224
    assert part != full
    hits = coverage(part, 2, 3840, 0, 2160)
    assert (hits == 1).all()


def test_plan_arguments(rmd):
    out = (C.c_int * 8)()
    for bad in ((0, 10, 0, 10, 0, 256), (10, 10, 5, 5, 0, 256), (10, 10, 0, 11, 0, 256), (10, 10, 0, 10, 5, 256), (10, 10, 0, 10, 0, 0)):
        assert rmd.lib.rmd_debug_atrous_plan(*bad, out) != 0
    assert rmd.lib.rmd_debug_atrous_plan(10, 10, 0, 10, 0, 256, None) != 0
