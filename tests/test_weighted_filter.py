"""FilterParams::type GAUSSIAN / CROSS / WAVELET on the uchar4 planes.

PARITY UNPINNED BY THE REFERENCE: it declares these modes (include/filter.cuh:12-19) and
implements none (every kernel uses w = 1, src/filter.cu:41,127).  The oracle
(oracle/box_oracle.c:orc_weighted_filter) states this build's semantics; the HIP kernel must match
it to +-1 LSB (float weights through expf vs v_exp_f32, then a truncating cast), with all but a
vanishing fraction of the bytes identical."""
import numpy as np
import pytest


def make_params(rmd, mode, **kw):
    base = dict(type=mode, depth=1, level=0, radius=2, sigmaSpace=1.5, sigmaColor=30.0, sigmaAlbedo=20.0, sigmaNormal=40.0)
    base.update(kw)
    return rmd.FilterParams(**base)


# ------------------------------------------------------------------------------- oracle, CPU
def test_oracle_gaussian_properties(rmd, orc):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (40, 56, 4), dtype=np.uint8)
    flat = np.full((20, 30, 4), 93, np.uint8)
    p = make_params(rmd, rmd.FilterParams.GAUSSIAN)
    out = orc.weighted_filter(flat, p)
    assert (out[..., :3] >= 92).all() and (out[..., :3] <= 93).all() and (out[..., 3] == 0).all()
    wide = make_params(rmd, rmd.FilterParams.GAUSSIAN, sigmaSpace=1e4)     # ~uniform weights = the box filter
    box = orc.box_filter(img, 2, 1, False)
    assert np.abs(orc.weighted_filter(img, wide).astype(int) - box.astype(int))[..., :3].max() <= 1


def test_oracle_cross_stops_at_edges(rmd, orc):
    img = np.zeros((16, 32, 4), np.uint8)
    img[:, :16, :3], img[:, 16:, :3] = 40, 200
    alb = img.copy()
    p = make_params(rmd, rmd.FilterParams.CROSS, sigmaColor=5.0, sigmaAlbedo=5.0, sigmaNormal=0.0)
    out = orc.weighted_filter(img, p, albedo=alb)
    assert np.abs(out[..., :3].astype(int) - img[..., :3].astype(int)).max() <= 1      # the step edge survives
    blur = orc.weighted_filter(img, make_params(rmd, rmd.FilterParams.GAUSSIAN))
    assert 41 < blur[8, 15, 0] < 199                                                     # a plain Gaussian smears it


def test_oracle_wavelet_levels_dilate(rmd, orc):
    img = np.zeros((64, 64, 4), np.uint8)
    img[32, 32, :3] = 255
    for level in range(3):
        p = make_params(rmd, rmd.FilterParams.WAVELET, level=level, sigmaColor=0.0, sigmaAlbedo=0.0, sigmaNormal=0.0)
        out = orc.weighted_filter(img, p)
        ys, xs = np.nonzero(out[..., 0])
        assert set(np.unique(ys - 32)) <= {d * (1 << level) for d in range(-2, 3)}      # taps sit on the dilated lattice
        assert out[32, 32, 0] == int(255 * 0.375 * 0.375)
    two = orc.weighted_filter(img, make_params(rmd, rmd.FilterParams.WAVELET, depth=2, sigmaColor=0.0, sigmaAlbedo=0.0, sigmaNormal=0.0))
    one = orc.weighted_filter(img, make_params(rmd, rmd.FilterParams.WAVELET, level=0, sigmaColor=0.0, sigmaAlbedo=0.0, sigmaNormal=0.0))
    again = orc.weighted_filter(one, make_params(rmd, rmd.FilterParams.WAVELET, level=1, sigmaColor=0.0, sigmaAlbedo=0.0, sigmaNormal=0.0))
    assert (two == again).all()                                                          # depth = repeated levels


# ------------------------------------------------------------------------------- the oracle against an independent restatement
def ref_weighted_f64(img, p, normal=None, albedo=None):
    """The three modes in plain 2-D order, float64, NO separability and NO fused multiply-adds: w = k * exp(-e) with
    GAUSSIAN  k = 1, e = (dx^2 + dy^2) / (2 sigmaSpace^2)                              over the (2r+1)^2 window
    CROSS     the same + |dc|^2 / (2 sigmaColor^2) + |da|^2 / (2 sigmaAlbedo^2) + |dn|^2 / (2 sigmaNormal^2)
    WAVELET   k = B3(dx) B3(dy) with B3 = {3/8, 1/4, 1/16} (reference src/filter.cu:10), taps at spacing 2^(level + l), e = the edge terms
    out = floor(sum w c / sum w) over the taps inside the frame (the reference's border rule, src/filter.cu:38-39,49), levels
    ping-pong as in src/filter.cu:24-25.  Written from include/rmd_api.h's description of the modes, not from the oracle's code:
    it pins what oracle/box_oracle.c MEANS, which its evaluation order (rewritten to the kernels' in round 3) cannot."""
    spline = (0.375, 0.25, 0.0625)
    mode, F = p.type, type(p)
    inv2 = lambda s: 1.0 / (2.0 * float(s) * float(s)) if s > 0 else 0.0          # noqa: E731
    is_s, is_c, is_a, is_n = inv2(p.sigmaSpace), inv2(p.sigmaColor), inv2(p.sigmaAlbedo), inv2(p.sigmaNormal)
    if mode == F.GAUSSIAN or not p.sigmaNormal > 0:
        normal = None
    if mode == F.GAUSSIAN or not p.sigmaAlbedo > 0:
        albedo = None
    cur = img[..., :3].astype(np.float64)
    H, W = cur.shape[:2]
    guides = [(g[..., :3].astype(np.float64), s) for g, s in ((albedo, is_a), (normal, is_n)) if g is not None]
    ys, xs = np.mgrid[0:H, 0:W]
    for level in range(p.depth):
        radius = 2 if mode == F.WAVELET else p.radius
        step = (1 << (p.level + level)) if mode == F.WAVELET else 1
        num, den = np.zeros((H, W, 3)), np.zeros((H, W))
        for dx in range(-radius, radius + 1):
            for dy in range(-radius, radius + 1):
                tx, ty = xs + dx * step, ys + dy * step
                ok = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
                txc, tyc = np.clip(tx, 0, W - 1), np.clip(ty, 0, H - 1)
                tap = cur[tyc, txc]
                k = spline[abs(dx)] * spline[abs(dy)] if mode == F.WAVELET else 1.0
                e = np.zeros((H, W)) if mode == F.WAVELET else np.full((H, W), (dx * dx + dy * dy) * is_s)
                if mode != F.GAUSSIAN:
                    e = e + ((cur - tap) ** 2).sum(-1) * is_c
                    for g, s in guides:
                        e = e + ((g - g[tyc, txc]) ** 2).sum(-1) * s
                w = np.where(ok, k * np.exp(-e), 0.0)
                num += w[..., None] * tap
                den += w
        quotient = num / den[..., None]
        cur = np.floor(quotient)
    out = np.zeros(img.shape, np.uint8)
    out[..., :3] = cur.astype(np.uint8)
    return out, quotient                   # (the last level's quotient before truncation)


@pytest.mark.parametrize("mode", ["GAUSSIAN", "CROSS", "WAVELET"])
def test_oracle_agrees_with_an_independent_float64_restatement(rmd, orc, mode):
    """Every byte of the oracle is the truncation of a quotient within fp32 rounding (1e-3 of a grey level) of the float64 one:
    wherever the float64 quotient is further than that from an integer the bytes are IDENTICAL, and elsewhere they are within
    1 LSB -- the oracle's evaluation order (separable GAUSSIAN, fused multiply-adds) may move a quotient across an integer,
    nothing more.  (How many quotients sit ON an integer depends on the picture, not on the implementation: wherever all the
    taps with weight are equal -- flat walls of the Cornell render, white noise where only the centre tap has weight -- the
    quotient is that value exactly and its last bit decides the byte: 1.7 % of the Cornell crop, a third of white noise.  On
    soft noise almost none, and there the usual < 0.5 % bound is asserted as well.)"""
    rng = np.random.default_rng(11)
    white = tuple(rng.integers(0, 256, (37, 61, 4), dtype=np.uint8) for _ in range(3))
    soft = tuple(((rng.integers(0, 256, (37, 61, 4), dtype=np.uint8) >> 2) + 96).astype(np.uint8) for _ in range(3))
    crop = tuple(orc.load_cornell(n)[180:340, 120:330].copy() for n in ("render", "normal", "albedo"))       # wall edge + box + noise
    cases = [dict(depth=1), dict(depth=1, radius=4, sigmaSpace=2.5, level=2), dict(depth=2, level=1)]
    for name, (render, normal, albedo) in (("white noise", white), ("soft noise", soft), ("cornell crop", crop)):
        for kw in cases:
            p = make_params(rmd, getattr(rmd.FilterParams, mode), **kw)
            want, q = ref_weighted_f64(render, p, normal, albedo)
            got = orc.weighted_filter(render, p, normal, albedo)
            assert (got[..., 3] == 0).all()
            g = got[..., :3].astype(np.float64)
            diff = np.abs(got[..., :3].astype(int) - want[..., :3].astype(int))
            assert diff.max() <= (1 if p.depth == 1 else 2), f"{mode} {kw} on {name}: a byte differs by {diff.max()}"
            if p.depth == 1:         # (with more levels the one-LSB choices of the first level feed the second: the bound above only)
                on_integer = np.abs(q - np.round(q)) < 1e-3
                assert (g == np.floor(q))[~on_integer].all(), f"{mode} {kw} on {name}: a byte away from a rounding boundary differs"
                assert ((g == np.floor(q + 1e-3)) | (g == np.floor(q - 1e-3))).all(), f"{mode} {kw} on {name}: a byte is not the truncation of the float64 quotient"
                if name == "soft noise":
                    assert (diff != 0).mean() < 5e-3, f"{mode} {kw} on {name}: {(diff != 0).mean():.4f} of the bytes differ"


def test_oracle_gaussian_takes_any_radius(rmd, orc):
    """(its table of weights used to be 64 entries read at |dx| up to the radius)"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (9, 150, 4), dtype=np.uint8)
    p = make_params(rmd, rmd.FilterParams.GAUSSIAN, radius=70, sigmaSpace=1e4)      # ~uniform weights over a window wider than the table was
    out = orc.weighted_filter(img, p)
    box = orc.box_filter(img, 70, 1, False)
    assert np.abs(out[..., :3].astype(int) - box[..., :3].astype(int)).max() <= 1


# ------------------------------------------------------------------------------- GPU parity
def gpu_run(rmd, img, p, normal=None, albedo=None):
    import torch
    t = torch.from_numpy(img).cuda()
    out = torch.empty_like(t)
    b0, b1 = torch.empty_like(t), torch.empty_like(t)
    g = rmd.make_gbuffer(t, out, b0, b1, normal=None if normal is None else torch.from_numpy(normal).cuda(),
                         albedo=None if albedo is None else torch.from_numpy(albedo).cuda())
    keep = (g, t, out, b0, b1)                      # noqa: F841  (planes stay alive through the launch)
    rmd.filterKernelTiled(g, p)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def assert_close_u8(got, want, exact=False):
    if exact:                # GAUSSIAN: no transcendental on the device, the order of the fp32 operations is the oracle's
        assert (got == want).all(), f"{(got != want).mean():.2e} of the bytes differ"
        return
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1, f"max byte difference {diff.max()}"
    assert (diff != 0).mean() < 2e-3, f"{(diff != 0).mean():.2e} of the bytes differ"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["GAUSSIAN", "CROSS", "WAVELET"])
def test_cornell_against_oracle(rmd, orc, cuda, mode):
    render, normal, albedo = (orc.load_cornell(n) for n in ("render", "normal", "albedo"))
    for kw in (dict(depth=1), dict(depth=3, level=0), dict(depth=1, radius=4, sigmaSpace=2.5, level=2)):
        p = make_params(rmd, getattr(rmd.FilterParams, mode), **kw)
        assert_close_u8(gpu_run(rmd, render, p, normal, albedo), orc.weighted_filter(render, p, normal, albedo), exact=mode == "GAUSSIAN")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (3, 7), (9, 200), (37, 61), (130, 257)])
def test_ragged_shapes_and_missing_planes(rmd, orc, cuda, shape):
    # (9, 200): the last 64 x 8 tile of the CROSS / WAVELET kernel holds one row and 8 columns; (1, 1), (3, 7): windows larger than the frame
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    alb = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    nrm = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    for mode in ("GAUSSIAN", "CROSS", "WAVELET"):
        p = make_params(rmd, getattr(rmd.FilterParams, mode), depth=2)
        assert_close_u8(gpu_run(rmd, img, p, None, alb), orc.weighted_filter(img, p, None, alb), exact=mode == "GAUSSIAN")   # no normal plane
        assert_close_u8(gpu_run(rmd, img, p, nrm, None), orc.weighted_filter(img, p, nrm, None), exact=mode == "GAUSSIAN")   # no albedo plane
        assert_close_u8(gpu_run(rmd, img, p, nrm, alb), orc.weighted_filter(img, p, nrm, alb), exact=mode == "GAUSSIAN")
        p0 = make_params(rmd, getattr(rmd.FilterParams, mode), sigmaAlbedo=0.0, sigmaNormal=0.0)
        assert_close_u8(gpu_run(rmd, img, p0), orc.weighted_filter(img, p0), exact=mode == "GAUSSIAN")
    for level in (3, 5):     # spacings 8 (the lattice tile), 16, 32 and 32, 64, 128 (the gather kernel)
        pw = make_params(rmd, rmd.FilterParams.WAVELET, level=level, depth=3)
        assert_close_u8(gpu_run(rmd, img, pw, nrm, alb), orc.weighted_filter(img, pw, nrm, alb))
    for radius in (0, 1, 3, 7, 12, 13, 20, 40):   # GAUSSIAN: the separable kernel (1..4 unrolled, up to 12 at run time), beyond it the direct kernel
        pg = make_params(rmd, rmd.FilterParams.GAUSSIAN, radius=radius, sigmaSpace=0.8 + radius)
        assert_close_u8(gpu_run(rmd, img, pg), orc.weighted_filter(img, pg), exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,level", [("CROSS", 0), ("WAVELET", 0), ("WAVELET", 1), ("WAVELET", 2), ("WAVELET", 3), ("WAVELET", 4)])
def test_4k_frame_against_the_oracle_on_row_bands(rmd, orc, cuda, mode, level):
    """The SHIPPED kernels at BASELINE's frame size (3840x2160: weighted_tile_kernel for CROSS and WAVELET spacings 1..8, the
    gather kernel for spacing 16) against the oracle on sampled row bands, full width: the top and bottom of the frame (border
    rule) and bands in the middle that straddle the kernel's 8-row tiles and its row lattices.  The oracle runs on a crop of the
    band + the rows its taps reach; where the crop's edge is not the frame's, only rows a full reach inside it are compared."""
    rng = np.random.default_rng(21 + level)
    H, W = 2160, 3840
    img = (rng.integers(0, 256, (H, W, 4), dtype=np.uint8) >> 2) + 96           # mid-range values: weights that are neither 0 nor 1
    nrm = (rng.integers(0, 256, (H, W, 4), dtype=np.uint8) >> 3) + 100
    alb = (rng.integers(0, 256, (H, W, 4), dtype=np.uint8) >> 3) + 100
    p = make_params(rmd, getattr(rmd.FilterParams, mode), depth=1, level=level)
    got = gpu_run(rmd, img, p, nrm, alb)
    reach = 2 * (1 << level) if mode == "WAVELET" else p.radius
    assert len(np.unique(got[..., :3])) > 8
    for y0, y1 in ((0, 20), (1075, 1099), (1620 - 3, 1620 + 13), (H - 20, H)):
        c0, c1 = max(0, y0 - reach), min(H, y1 + reach)
        want = orc.weighted_filter(img[c0:c1].copy(), p, nrm[c0:c1].copy(), alb[c0:c1].copy())
        assert_close_u8(got[y0:y1], want[y0 - c0:y1 - c0])


@pytest.mark.gpu
@pytest.mark.experiments
@pytest.mark.parametrize("mode", ["CROSS", "WAVELET"])
def test_tile_kernel_equals_gather_kernel_at_4k(rmd, cuda, mode, monkeypatch):
    """Size-independent property at BASELINE's frame size: the LDS-tile kernel (two pixels per thread, packed arithmetic,
    squared distances from packed differences of the staged floats) and the one-pixel-per-thread gather kernel state the same
    fp32 operations in the same order, so they agree byte for byte -- whatever v_exp_f32 returns.  (The gather kernel is
    selected by RMD_WEIGHTED_TILE=0, which only the experiments build reads.)"""
    rng = np.random.default_rng(21)
    shape = (2160, 3840, 4)
    img = (rng.integers(0, 256, shape, dtype=np.uint8) >> 2) + 96           # mid-range values: weights that are neither 0 nor 1
    nrm = (rng.integers(0, 256, shape, dtype=np.uint8) >> 3) + 100
    alb = (rng.integers(0, 256, shape, dtype=np.uint8) >> 3) + 100
    p = make_params(rmd, getattr(rmd.FilterParams, mode), depth=5 if mode == "WAVELET" else 1)         # WAVELET: spacings 1 .. 16
    tile = gpu_run(rmd, img, p, nrm, alb)
    monkeypatch.setenv("RMD_WEIGHTED_TILE", "0")
    gather = gpu_run(rmd, img, p, nrm, alb)
    assert len(np.unique(tile[..., :3])) > 8                               # neither saturated nor flat (five levels smooth noise to ~11 values)
    assert (tile == gather).all(), f"{(tile != gather).mean():.2e} of the bytes differ"


@pytest.mark.gpu
def test_parameter_errors(rmd, cuda):
    import torch
    t = torch.zeros((8, 8, 4), dtype=torch.uint8, device="cuda")
    g = rmd.make_gbuffer(t, torch.empty_like(t))
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, make_params(rmd, rmd.FilterParams.GAUSSIAN, sigmaSpace=0.0))
    assert e.value.code == -3
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, make_params(rmd, rmd.FilterParams.WAVELET, level=12))
    assert e.value.code == -3
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, make_params(rmd, rmd.FilterParams.GAUSSIAN, radius=128))
    assert e.value.code == -3
