"""uchar4 filter parity on the GPU: HIP kernels (through the C ABI) vs the oracle and the
SURVEY §8(c) known answers.  Bit-exact (RGB; .w is 0 here, uninitialised in the reference
baseline, src/filter.cu:50)."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_box import KNOWN  # noqa: E402


def run_gpu(rmd, img, radius=2, depth=1, tiled=True, cache_input=True):
    t = torch.from_numpy(img).cuda()
    p = rmd.FilterParams(type=rmd.FilterParams.AVERAGE, depth=depth, radius=radius, cacheInput=cache_input)
    out = rmd.box_filter(t, p, tiled=tiled)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("size", ["500", "256"])
@pytest.mark.parametrize("kind,cache", [("baseline", True), ("tiled", False), ("tiled", True)])
def test_cornell_known_answers(rmd, orc, cuda, size, kind, cache):
    """BASELINE config 1 on its GPU twin.  cacheInput=true (LDS path) must equal the reference's
    cacheInput=false result: the reference's own cached path has a stride bug."""
    full = orc.load_cornell("render")
    img = full if size == "500" else full[122:378, 122:378].copy()
    out = run_gpu(rmd, img, 2, 1, tiled=(kind == "tiled"), cache_input=cache)
    assert hashlib.sha256(np.ascontiguousarray(out[:, :, :3]).tobytes()).hexdigest() == KNOWN[(size, kind)]
    assert (out[:, :, 3] == 0).all()


@pytest.mark.parametrize("shape,r", [((1, 1), 2), ((3, 7), 0), ((5, 4), 3), ((37, 61), 2), ((16, 16), 9), ((130, 257), 5),
                                     ((70, 66), 24), ((40, 40), 30)])
@pytest.mark.parametrize("tiled,cache", [(False, True), (True, False), (True, True)])
def test_edge_shapes(rmd, orc, cuda, shape, r, tiled, cache):
    """Empty-ish and ragged inputs: 1x1, radius 0, radius larger than the image, sizes that are not
    multiples of a tile, radii the scan kernel takes (up to 32: 3, 5, 9, 24, 30 here) and radius 0 (direct kernel)."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    got = run_gpu(rmd, img, r, 1, tiled, cache)
    want = orc.box_filter(img, r, 1, gray_from_r=not tiled)
    assert (got == want).all()


@pytest.mark.parametrize("r", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(1, 4), (3, 8), (20, 252), (33, 256), (40, 260), (75, 512), (130, 1028), (7, 2052)])
def test_stream_kernel_radii_and_ragged_strips(rmd, orc, cuda, shape, r):
    """The radius 1..4 fast path (widths that are multiples of 4: 256-pixel strips walked by one wave,
    neighbours through DPP wave shifts, strip halos from lanes 0 / 63, 16-row bands with a register
    ring): one strip, strips that end in the middle of a wave, a last strip of a single lane, frames
    shorter than the window and than a band; RGB (tiled) and gray-from-R (baseline), bit-exact."""
    rng = np.random.default_rng(100 * r + shape[1])
    img = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    img[rng.integers(0, shape[0]), :, :] = 255                     # a saturated row: the largest sums
    for tiled in (True, False):
        got = run_gpu(rmd, img, r, 1, tiled, True)
        want = orc.box_filter(img, r, 1, gray_from_r=not tiled)
        assert (got == want).all(), (shape, r, tiled, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("r", [1, 5, 8, 16, 31, 32, 33])
@pytest.mark.parametrize("shape", [(70, 61), (100, 200), (33, 130)])
def test_scan_kernel_radii(rmd, orc, cuda, shape, r):
    """Radii 1..32 outside the stream kernel's reach (here: widths that are not multiples of 4, or radius > 4) run prefix sums
    along the rows and running sums down the columns; 33 is the first radius of the run kernel.  A saturated frame makes
    the largest sums the packed 16-bit prefix fields and the reciprocal division have to carry; bit-exact either way."""
    rng = np.random.default_rng(7 * r + shape[1])
    for img in (rng.integers(0, 256, shape + (4,), dtype=np.uint8), np.full(shape + (4,), 255, np.uint8)):
        for tiled, cache in ((True, True), (True, False), (False, True)):
            got = run_gpu(rmd, img, r, 1, tiled, cache)
            want = orc.box_filter(img, r, 1, gray_from_r=not tiled)
            assert (got == want).all(), (shape, r, tiled, cache, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("r", [33, 40, 64, 100, 127, 128, 140])
@pytest.mark.parametrize("shape", [(70, 61), (300, 200), (260, 515)])
def test_run_kernel_radii(rmd, orc, cuda, shape, r):
    """Radii 33 .. 127 (the reference takes any radius, src/filter.cu:34) run box_run_kernel: chunked prefix sums along the
    rows, a running sum down a wave's band of rows, the reference's own float division.  Window sums stay below 2^24 there, so
    they equal the reference's float accumulation in any order; from radius 128 on they do not, and the direct kernel, which
    follows the reference's tap order, takes over (128, 140 here).  Windows larger than the frame, ragged strips, saturated
    frames (the largest sums); bit-exact."""
    rng = np.random.default_rng(3 * r + shape[1])
    for img in (rng.integers(0, 256, shape + (4,), dtype=np.uint8), np.full(shape + (4,), 255, np.uint8)):
        for tiled, cache in ((True, True), (False, False)):
            got = run_gpu(rmd, img, r, 1, tiled, cache)
            want = orc.box_filter(img, r, 1, gray_from_r=not tiled, threads=8)
            assert (got == want).all(), (shape, r, tiled, cache, np.argwhere(got != want)[:4])


def test_run_kernel_on_a_wide_frame(rmd, orc, cuda):
    """Radius 64 on half of the reference's bench shape (1920 x 540): many bands per column strip, 30 strips; against the oracle
    on all host threads."""
    rng = np.random.default_rng(64)
    img = rng.integers(0, 256, (540, 1920, 4), dtype=np.uint8)
    got = run_gpu(rmd, img, 64, 1, True, True)
    want = orc.box_filter(img, 64, 1, gray_from_r=False, threads=16)
    assert (got == want).all()


@pytest.mark.parametrize("tiled", [False, True])
def test_multi_level_ping_pong(rmd, orc, cuda, tiled):
    """depth>1 with the reference's plane routing (src/filter.cu:24-25), one launch per level."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (90, 150, 4), dtype=np.uint8)
    for depth in (2, 3, 4):
        assert (run_gpu(rmd, img, 2, depth, tiled) == orc.box_filter(img, 2, depth, gray_from_r=not tiled)).all()


def test_1080p_full_size(rmd, orc, cuda):
    """The reference's bench shape (src/test.cu:64-90): 1920x1080, radius 2, depth 1."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8)
    want = orc.box_filter(img, 2, 1, False, threads=8)
    for cache in (False, True):
        assert (run_gpu(rmd, img, 2, 1, True, cache) == want).all()
    assert (run_gpu(rmd, img, 2, 1, False)[:, :, 0] == want[:, :, 0]).all()


def test_4k_properties(rmd, cuda):
    """Size-independent properties at 3840x2160: a constant image is a fixed point, the filter is
    monotone, and LDS / direct paths agree."""
    t = torch.full((2160, 3840, 4), 77, dtype=torch.uint8, device="cuda")
    p = rmd.FilterParams(radius=2)
    assert (rmd.box_filter(t, p)[:, :, :3] == 77).all()
    rng = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randint(0, 256, (2160, 3840, 4), dtype=torch.uint8, device="cuda", generator=rng)
    a = rmd.box_filter(img, rmd.FilterParams(radius=3, cacheInput=True))
    b = rmd.box_filter(img, rmd.FilterParams(radius=3, cacheInput=False))
    assert torch.equal(a, b)
    brighter = torch.clamp(img.to(torch.int16) + 9, max=255).to(torch.uint8)
    c = rmd.box_filter(brighter, rmd.FilterParams(radius=3))
    assert (c[:, :, :3] >= a[:, :, :3]).all()


def test_errors_surface_as_exceptions(rmd, cuda):
    t = torch.zeros((8, 8, 4), dtype=torch.uint8, device="cuda")
    g = rmd.make_gbuffer(t, torch.empty_like(t))
    with pytest.raises(rmd.RmdError):
        rmd.filterKernelTiled(g, rmd.FilterParams(depth=2))          # buffer[] missing
    g2 = rmd.make_gbuffer(t, t.clone())
    g2.denoised = g2.render
    with pytest.raises(rmd.RmdError):
        rmd.filterKernelBaseline(g2, rmd.FilterParams())              # aliasing
