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
        for k in ("t_color", "t_moments", "t_len", "t_debug", "v_color", "hist_color_out", "out_color"):
            out[f"f{i}_{k}"] = getattr(fr, k).copy()
        hc, hm, pn = fr.history()
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
        hc, hm, pn = fr.t_color, (fr.t_moments, fr.t_len), nd


def test_temporal_rejects_out_of_frame_and_far_taps(orc):
    h, w = 12, 12
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 10
    c = np.ones((h, w, 4), np.float32)
    hist = np.ones((h, w, 4), np.float32)
    hist_m = (np.ones((h, w, 2), np.float32), np.full((h, w), 5, np.uint8))          # moments (m1, m2) and history length 5
    p = orc.default_params()
    m = np.zeros((h, w, 2), np.float32); m[..., 0] = -3.0       # reproject 3 px to the left
    fr = orc.Frame(w, h, c, nd, m, hist, hist_m, nd)
    orc.temporal(fr, p)
    assert (fr.t_debug[:, :3, 3] == 1).all() and (fr.t_debug[:, 3:, 3] == 6).all()
    p.max_motion_rows = 2
    m2 = np.zeros((h, w, 2), np.float32); m2[..., 1] = 4.0      # 4 rows down: beyond max_motion_rows
    fr = orc.Frame(w, h, c, nd, m2, hist, hist_m, nd)
    orc.temporal(fr, p)
    assert (fr.t_debug[..., 2] == 0).all() and (fr.t_debug[..., 3] == 1).all()


def test_variance_passes_long_history_through(orc):
    h, w = 10, 14
    rng = np.random.default_rng(1)
    c = rng.random((h, w, 4), dtype=np.float32)
    nd = np.zeros((h, w, 4), np.float32); nd[..., 2] = 1; nd[..., 3] = 3
    fr = orc.Frame(w, h, c, nd, np.zeros((h, w, 2), np.float32))
    fr.t_color[:] = c
    fr.t_len[...] = 4
    fr.t_len[:, :5] = 2
    orc.variance(fr, orc.default_params())
    assert (fr.v_color[:, 5:] == c[:, 5:]).all()
    assert not (fr.v_color[:, :5] == c[:, :5]).all()
    assert (fr.v_color[..., 3] >= 0).all()


def test_frame_threads_do_not_change_result(orc):
    ins = orc.synth_gbuffer(64, 48, 1)
    a = orc.Frame(64, 48, *ins); b = orc.Frame(64, 48, *ins)
    orc.frame(a, orc.default_params(), 1); orc.frame(b, orc.default_params(), 4)
    assert (a.out_color == b.out_color).all() and (a.hist_color_out == b.hist_color_out).all()


# ---- hand-computed known answers (SURVEY Appendix A evaluated with pencil-and-paper numpy, independent of
# ---- oracle/svgf_oracle.c: the goldens above only guard the oracle against drift, these pin its meaning) ----
LUM = np.array([0.2126, 0.7152, 0.0722], np.float64)


def test_temporal_blend_weights_by_hand(orc):
    """T on a 1x1 frame, static camera.  Frame 0 has no history: h = 1, alpha = 1, variance m2 - m1^2 = 0.
    Frame 1 re-projects onto itself: h = 2, alpha = max(0.05, 1/2), alpha_m = max(0.2, 1/2) (A.T.3-4)."""
    p = orc.default_params()
    nd = np.array([[[0.0, 0.0, 1.0, 5.0]]], np.float32)
    m = np.zeros((1, 1, 2), np.float32)
    c0 = np.array([[[1.0, 2.0, 4.0, 0.0]]], np.float32)
    f0 = orc.Frame(1, 1, c0, nd, m)
    orc.temporal(f0, p)
    l0 = float(LUM @ c0[0, 0, :3])
    assert np.allclose(f0.t_color[0, 0], [1, 2, 4, 0], atol=1e-6) and f0.t_debug[0, 0, 3] == 1
    assert np.allclose(f0.t_moments[0, 0], [l0, l0 * l0], rtol=1e-6) and f0.t_len[0, 0] == 1
    c1 = np.array([[[3.0, 2.0, 0.0, 0.0]]], np.float32)
    f1 = orc.Frame(1, 1, c1, nd, m, f0.t_color, (f0.t_moments, f0.t_len), nd)
    orc.temporal(f1, p)
    l1 = float(LUM @ c1[0, 0, :3])
    m1 = 0.5 * l0 + 0.5 * l1
    m2 = 0.5 * l0 * l0 + 0.5 * l1 * l1
    assert f1.t_debug[0, 0].tolist() == [0, 0, 1, 2]              # q0 = (0,0); only tap (0,0) is inside the frame; h = 2
    assert np.allclose(f1.t_color[0, 0, :3], [2.0, 2.0, 2.0], rtol=1e-6)
    assert abs(f1.t_color[0, 0, 3] - (m2 - m1 * m1)) <= 2e-6 * m2          # fp32 cancellation of two values near 4
    assert np.allclose(f1.t_moments[0, 0], [m1, m2], rtol=1e-6) and f1.t_len[0, 0] == 2
    # third frame: h = 3, alpha = 1/3, alpha_m = 1/3
    f2 = orc.Frame(1, 1, c0, nd, m, f1.t_color, (f1.t_moments, f1.t_len), nd)
    orc.temporal(f2, p)
    assert f2.t_debug[0, 0, 3] == 3
    assert np.allclose(f2.t_color[0, 0, :3], (2.0 * np.array([2, 2, 2]) + np.array([1, 2, 4])) / 3.0, rtol=1e-6)


def test_temporal_rejects_history_across_a_depth_or_normal_edge_by_hand(orc):
    p = orc.default_params()
    m = np.zeros((1, 1, 2), np.float32)
    c = np.array([[[1.0, 1.0, 1.0, 0.0]]], np.float32)
    nd_prev = np.array([[[0.0, 0.0, 1.0, 5.0]]], np.float32)
    hist_m = (np.array([[[1.0, 1.0]]], np.float32), np.array([[7]], np.uint8))          # (m1, m2), history length 7
    for nd_cur, ok in (([0, 0, 1, 5.05], True),      # |dz| = 0.05 <= k_z (g_z + 1e-2) = 0.1
                       ([0, 0, 1, 5.2], False),      # |dz| = 0.2 > 0.1
                       ([0, 1, 0, 5.0], False)):     # n.n' = 0 < k_n = 0.9
        f = orc.Frame(1, 1, c, np.array([[nd_cur]], np.float32), m, c, hist_m, nd_prev)
        orc.temporal(f, p)
        assert f.t_debug[0, 0, 3] == (8 if ok else 1), nd_cur      # h = min(32, 7 + 1) or a disocclusion


def test_atrous_weights_by_hand(orc):
    """A on a 3x1 frame, iteration 0: the centre pixel sees taps dx = -1, 0, +1 (dy = 0), everything else is out of
    frame and skipped.  k = 3/8 * {1/4, 3/8, 1/4}; same normal (w_n = 1); z differs on the right tap only."""
    p = orc.default_params()
    nd = np.zeros((1, 3, 4), np.float32)
    nd[..., 2] = 1.0
    nd[0, :, 3] = [2.0, 2.0, 2.5]
    col = np.zeros((1, 3, 4), np.float32)
    col[0, :, :3] = [[1, 1, 1], [1, 1, 1], [3, 3, 3]]
    col[0, :, 3] = [0.04, 0.04, 0.04]
    f = orc.Frame(3, 1, col, nd, np.zeros((1, 3, 2), np.float32))
    out = np.zeros_like(col)
    orc.atrous(f, p, 0, col, out)
    # centre pixel x = 1: g_z = |z(2) - z(1)| + |z(1, y+1 clamped) - z(1)| = 0.5; var_c: 3x1 prefilter, weights 1/8 1/4 1/8
    var_c = 0.04
    k = np.array([0.25, 0.375, 0.25]) * 0.375
    lum = np.array([1.0, 1.0, 3.0])
    w_z = np.array([0.0, 0.0, 0.5 / (1.0 * 0.5 * 1 * 1.0 + 1e-8)])
    w_l = np.abs(lum[1] - lum) / (4.0 * np.sqrt(var_c) + 1e-8)
    w = k * np.exp(-w_z - w_l)
    want_c = (w * lum).sum() / w.sum()
    want_v = (w * w * 0.04).sum() / w.sum() ** 2
    assert np.allclose(out[0, 1, :3], want_c, rtol=1e-5)
    assert np.isclose(out[0, 1, 3], want_v, rtol=1e-5)
    # a perpendicular normal on the right tap removes it entirely: w_n = max(0, 0)^128 = 0
    nd2 = nd.copy()
    nd2[0, 2, :3] = [1.0, 0.0, 0.0]
    f2 = orc.Frame(3, 1, col, nd2, np.zeros((1, 3, 2), np.float32))
    orc.atrous(f2, p, 0, col, out)
    assert np.allclose(out[0, 1, :3], 1.0, rtol=1e-6)


def test_atrous_single_pixel_frame_is_the_identity(orc):
    p = orc.default_params()
    nd = np.array([[[0.0, 0.6, 0.8, 3.0]]], np.float32)
    col = np.array([[[0.3, 0.7, 0.1, 0.25]]], np.float32)
    f = orc.Frame(1, 1, col, nd, np.zeros((1, 1, 2), np.float32))
    src = col
    for it in range(5):
        out = np.zeros_like(col)
        orc.atrous(f, p, it, src, out)
        assert np.allclose(out, col, rtol=1e-6), it          # only the centre tap exists: c k / k, var k^2 / k^2
        src = out


def test_demodulation_by_hand(orc):
    rad = np.array([[[0.5, 0.2, 0.0, 0.7], [1.0, 1.0, 1.0, 0.1]]], np.float32)
    alb = np.array([[[0.25, 0.8, 0.0, 1.0], [0.5, 0.0005, 2.0, 1.0]]], np.float32)
    got = orc.demodulate(rad, alb, eps=1e-3)
    want = np.array([[[2.0, 0.25, 0.0, 0.7], [2.0, 1000.0, 0.5, 0.1]]], np.float32)
    assert np.allclose(got, want, rtol=1e-6)


def test_variance_spatial_estimate_by_hand(orc):
    """V on a 3x3 frame (Appendix A.V): the centre pixel has h = 2 < 4, so it takes the 7x7 spatial estimate -- of which
    only the 3x3 frame exists (out-of-frame taps skipped and renormalised).  Flat plane (same normal, w_n = 1), depth
    varies along x only: g_z(centre) = |z(2,1) - z(1,1)| + |z(1,2) - z(1,1)| = 0.25; weights w = exp(-|dz| / (sigma_z
    g_z len + 1e-8)), no luminance term.  variance = max(0, E[l^2] - E[l]^2) * 4 / h; colour = weighted mean.
    A pixel with h = 4 passes through untouched."""
    p = orc.default_params()
    nd = np.zeros((3, 3, 4), np.float32)
    nd[..., 2] = 1.0
    nd[..., 3] = np.array([4.0, 4.25, 4.5], np.float32)[None, :]          # z depends on x only
    t_color = np.zeros((3, 3, 4), np.float32)
    grey = np.array([[1.0, 2.0, 4.0], [0.5, 1.0, 3.0], [2.0, 2.0, 0.25]], np.float64)
    t_color[..., :3] = grey[..., None].astype(np.float32) * np.array([1.0, 0.5, 2.0], np.float32)   # r, g, b = l', l'/2, 2 l'
    t_color[..., 3] = 0.125
    t_len = np.full((3, 3), 4, np.uint8)                                   # everybody has a long history ...
    t_len[1, 1] = 2                                                        # ... except the centre
    f = orc.Frame(3, 3, t_color, nd, np.zeros((3, 3, 2), np.float32))
    f.t_color[...] = t_color
    f.t_len[...] = t_len
    orc.variance(f, p)
    # by hand, in float64
    gz, zc = 0.25, 4.25
    sw = sl = sl2 = 0.0
    sc = np.zeros(3)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            x, y = 1 + dx, 1 + dy
            length = np.sqrt(dx * dx + dy * dy)
            wz = abs(zc - float(nd[y, x, 3])) / (1.0 * gz * length + 1e-8) if (dx or dy) else 0.0
            w = np.exp(-wz)
            rgb = t_color[y, x, :3].astype(np.float64)
            lum = float(LUM @ rgb)
            sw += w; sc += w * rgb; sl += w * lum; sl2 += w * lum * lum
    want_var = max(0.0, sl2 / sw - (sl / sw) ** 2) * 4.0 / 2.0
    assert np.allclose(f.v_color[1, 1, :3], sc / sw, rtol=1e-5)
    assert np.isclose(f.v_color[1, 1, 3], want_var, rtol=1e-4)
    # the two taps on the centre column (dx = 0) have dz = 0 and weight exactly 1; the others exp(-0.25 / (0.25 len)):
    # exp(-1) beside the centre, exp(-1/sqrt 2) on the diagonals
    assert np.isclose(sw, 3.0 + 2.0 * np.exp(-1.0) + 4.0 * np.exp(-1.0 / np.sqrt(2.0)), rtol=1e-6)
    others = np.ones((3, 3), bool)
    others[1, 1] = False
    assert (f.v_color[others] == t_color[others]).all()                   # h >= var_h_threshold: pass-through, bit for bit
