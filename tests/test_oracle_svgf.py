"""Oracle pinning, SVGF passes: committed golden vectors + spec properties.  CPU only.

PARITY UNPINNED BY THE REFERENCE (no SVGF code or vectors exist there): the
goldens were produced by this oracle (tests/golden/make_golden.py) and guard it against drift.
"""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def golden(orc):
    return np.load(os.path.join(orc.ROOT, "tests", "golden", "svgf_golden.npz"))


def run_sequence(orc, width, height, inputs, p):
    out, hc, hm, pn = {}, None, None, None
    for i, (c, nd, m) in enumerate(inputs):
        fr = orc.Frame(width, height, c, nd, m, hc, hm, pn)
        orc.frame(fr, p)
        for k in ("t_color", "t_moments", "t_debug", "v_color", "hist_color_out", "out_color"):
            out[f"f{i}_{k}"] = getattr(fr, k).copy()
        hc, hm, pn = fr.hist_color_out, fr.t_moments, fr.nd
    return out


def test_synthetic_sequence_matches_golden(orc, golden):
    p = orc.default_params()
    res = run_sequence(orc, 64, 48, [orc.synth_gbuffer(64, 48, f) for f in range(3)], p)
    for k, v in res.items():
        g = golden["synth_" + k]
        if v.dtype == np.int32:
            assert (v == g).all(), k                      # q0 / mask / h are bit-exact outputs
        else:
            np.testing.assert_array_equal(v, g, err_msg=k)  # same binary, same libm: exact


def test_cornell_crop_matches_golden(orc, golden):
    p = orc.default_params()
    c, nd, m = orc.cornell_svgf_inputs()
    crop = (slice(200, 280), slice(150, 246))
    c, nd, m = c[crop].copy(), nd[crop].copy(), m[crop].copy()
    res = run_sequence(orc, 96, 80, [(c, nd, m)] * 2, p)
    for k, v in res.items():
        np.testing.assert_array_equal(v, golden["cornell_" + k], err_msg=k)


def test_synth_is_deterministic_and_in_range(orc, golden):
    c, nd, m = orc.synth_gbuffer(64, 48, 0)
    assert float(c.astype(np.float64).sum()) == float(golden["synth_color0_sum"][0])
    assert c.min() >= 0 and c.max() <= 16 and (nd[..., 3] >= 1).all() and (nd[..., 3] <= 100).all()
    n2 = (nd[..., :3] ** 2).sum(-1)
    assert np.allclose(n2, 1, atol=1e-5)
    # strips generate the same pixels as the whole frame
    cs, nds, ms = orc.synth_gbuffer(64, 48, 0, buf_row0=10, buf_rows=20)
    assert (cs == c[10:30]).all() and (nds == nd[10:30]).all() and (ms == m[10:30]).all()


def test_atrous_constant_image_is_a_fixed_point(orc):
    h, w = 24, 40
    c = np.zeros((h, w, 4), np.float32); c[..., :3] = (0.25, 0.5, 0.75); c[..., 3] = 0.1
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 5
    fr = orc.Frame(w, h, c, nd, np.zeros((h, w, 2), np.float32))
    p = orc.default_params()
    for it in range(5):
        out = np.zeros_like(c)
        orc.atrous(fr, p, it, c, out)
        assert np.allclose(out[..., :3], c[..., :3], rtol=1e-6)
        assert (out[..., 3] <= c[..., 3] + 1e-7).all()          # variance never grows under averaging


def test_atrous_zero_normal_rules(orc):
    """Both normals zero => weight 1, exactly one zero => 0 (Appendix A.A.2)."""
    h, w = 9, 9
    c = np.zeros((h, w, 4), np.float32); c[..., :3] = 1.0; c[4, 4, :3] = 3.0
    nd = np.zeros((h, w, 4), np.float32); nd[..., 3] = 1.0
    nd[4, 4, :3] = (0, 0, 1)                                   # a single surface pixel among background
    fr = orc.Frame(w, h, c, nd, np.zeros((h, w, 2), np.float32))
    out = np.zeros_like(c)
    orc.atrous(fr, orc.default_params(), 0, c, out)
    assert np.allclose(out[4, 4, :3], 3.0)                       # only its own tap counts
    assert np.allclose(out[3, 3, :3], 1.0)                       # background ignores the surface pixel


def test_temporal_static_scene_accumulates(orc):
    h, w = 16, 16
    rng = np.random.default_rng(0)
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 10
    m = np.zeros((h, w, 2), np.float32)
    p = orc.default_params()
    hc = hm = pn = None
    for f in range(6):
        c = np.zeros((h, w, 4), np.float32); c[..., :3] = 1 + 0.5 * rng.standard_normal((h, w, 1)).astype(np.float32)
        fr = orc.Frame(w, h, c, nd, m, hc, hm, pn)
        orc.temporal(fr, p)
        assert (fr.t_debug[..., 3] == min(f + 1, p.h_max)).all()
        assert (fr.t_debug[:-1, :-1, 2] == (0 if f == 0 else 15)).all()   # last row/col: +1 taps leave the frame
        if f: assert (fr.t_debug[:-1, -1, 2] == 5).all() and (fr.t_debug[-1, :-1, 2] == 3).all() and fr.t_debug[-1, -1, 2] == 1
        assert (fr.t_debug[..., 0] == np.arange(w)[None, :]).all() and (fr.t_debug[..., 1] == np.arange(h)[:, None]).all()
        hc, hm, pn = fr.t_color, fr.t_moments, nd


def test_temporal_rejects_out_of_frame_and_far_taps(orc):
    h, w = 12, 12
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 10
    c = np.ones((h, w, 4), np.float32)
    hist = np.ones((h, w, 4), np.float32); hist[..., 2] = 5
    p = orc.default_params()
    m = np.zeros((h, w, 2), np.float32); m[..., 0] = -3.0       # reproject 3 px to the left
    fr = orc.Frame(w, h, c, nd, m, hist, hist, nd)
    orc.temporal(fr, p)
    assert (fr.t_debug[:, :3, 3] == 1).all() and (fr.t_debug[:, 3:, 3] == 6).all()
    p.max_motion_rows = 2
    m2 = np.zeros((h, w, 2), np.float32); m2[..., 1] = 4.0      # 4 rows down: beyond max_motion_rows
    fr = orc.Frame(w, h, c, nd, m2, hist, hist, nd)
    orc.temporal(fr, p)
    assert (fr.t_debug[..., 2] == 0).all() and (fr.t_debug[..., 3] == 1).all()


def test_variance_passes_long_history_through(orc):
    h, w = 10, 14
    rng = np.random.default_rng(1)
    c = rng.random((h, w, 4), dtype=np.float32)
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 3
    fr = orc.Frame(w, h, c, nd, np.zeros((h, w, 2), np.float32))
    fr.t_color[:] = c
    fr.t_moments[..., 2] = 4
    fr.t_moments[:, :5, 2] = 2
    orc.variance(fr, orc.default_params())
    assert (fr.v_color[:, 5:] == c[:, 5:]).all()
    assert not (fr.v_color[:, :5] == c[:, :5]).all()
    assert (fr.v_color[..., 3] >= 0).all()


def test_frame_threads_do_not_change_result(orc):
    ins = orc.synth_gbuffer(64, 48, 1)
    a = orc.Frame(64, 48, *ins); b = orc.Frame(64, 48, *ins)
    orc.frame(a, orc.default_params(), 1); orc.frame(b, orc.default_params(), 4)
    assert (a.out_color == b.out_color).all() and (a.hist_color_out == b.hist_color_out).all()
