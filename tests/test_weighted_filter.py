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
    for radius in (0, 1, 3, 7, 12):          # GAUSSIAN: every radius the separable kernel takes (1..4 unrolled, the rest at run time)
        pg = make_params(rmd, rmd.FilterParams.GAUSSIAN, radius=radius, sigmaSpace=0.8 + radius)
        assert_close_u8(gpu_run(rmd, img, pg), orc.weighted_filter(img, pg), exact=True)


@pytest.mark.gpu
@pytest.mark.experiments
@pytest.mark.parametrize("mode", ["CROSS", "WAVELET"])
def test_tile_kernel_equals_gather_kernel_at_4k(rmd, cuda, mode, monkeypatch):
    """Size-independent property at BASELINE's frame size: the LDS-tile kernel (two pixels per thread, packed arithmetic,
    distances as |k|^2 + |t|^2 - 2 k.t) and the one-pixel-per-thread gather kernel state the same fp32 operations in the same
    order, so they agree byte for byte -- whatever v_exp_f32 returns.  (The gather kernel is selected by RMD_WEIGHTED_TILE=0,
    which only the experiments build reads.)"""
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
        rmd.filterKernelTiled(g, make_params(rmd, rmd.FilterParams.GAUSSIAN, radius=13))
    assert e.value.code == -3
