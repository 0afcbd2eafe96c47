"""Oracle pinning, uchar4 box filter: SURVEY §8(c) known answers + edge cases.  CPU only."""
import hashlib

import numpy as np
import pytest

# SHA-256 over tightly packed RGB bytes of `denoised`, radius=2, depth=1 (derived
# there from the reference's own kernel source, src/filter.cu, on render/cornell/1/render.png).
KNOWN = {
    ("500", "baseline"): "b42c68daf74304b4b6f3f2f6314e11e2856c3f295a07a989e0a3a8627e444ac5",
    ("500", "tiled"): "1aae238680a4bc8e1ed66fda890e5978ac5b15ef5b11c63e34c7c029b310370b",
    ("256", "baseline"): "8c28277365bba7b0b7e59ddd750461ffbb5f24b5417d0c1b7e9beebc47cdd362",
    ("256", "tiled"): "2d3d222f281644c97143424166b870e81202d3607ce6d12c26bac8ee40e8f950",
}
FIXTURE_SHA = {
    "render": "070dc918aea79244aa61817c397d853f864c87a41d0672c2b488b4d2b2f1dc29",
    "albedo": "0c9601f2625f9298d13d436287c28faa32e2e2dd034d58ffde96a8ef39d3332b",
    "normal": "f7f929bef5f276561731a9ea38daa95a1aca85b23fe5b96e45e38124ad7241c3",
    "depth": "7888cdb5b934bbe916592052d3319f0f7e5f4b871558e390d0586578a5c62ee1",
}
RGBA_SHA = "e6bc2e8029fe4d0edbc4e0387ba30819350ee5ba3e4ed28d667e8ccf1e89c93e"


def rgb_sha(img):
    return hashlib.sha256(np.ascontiguousarray(img[:, :, :3]).tobytes()).hexdigest()


def cornell_inputs(orc):
    full = orc.load_cornell("render")
    return {"500": full, "256": full[122:378, 122:378].copy()}


def test_fixture_files_are_the_reference_ones(orc):
    import os
    for name, sha in FIXTURE_SHA.items():
        path = os.path.join(orc.ROOT, "tests", "golden", "cornell", f"{name}.png")
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == sha
    assert hashlib.sha256(orc.load_cornell("render").tobytes()).hexdigest() == RGBA_SHA


@pytest.mark.parametrize("size", ["500", "256"])
@pytest.mark.parametrize("kind", ["baseline", "tiled"])
def test_known_answers(orc, size, kind):
    out = orc.box_filter(cornell_inputs(orc)[size], radius=2, depth=1, gray_from_r=(kind == "baseline"))
    assert rgb_sha(out) == KNOWN[(size, kind)]


def test_known_answer_statistics(orc):
    out = orc.box_filter(orc.load_cornell("render"), 2, 1, False)
    assert out[:, :, :3].reshape(-1, 3).sum(0).tolist() == [22249763, 22806327, 14113950]
    assert out[250, 250, :3].tolist() == [127, 135, 126]
    gray = orc.box_filter(orc.load_cornell("render"), 2, 1, True)
    assert gray[250, 250, :3].tolist() == [127, 127, 127]
    assert (gray[:, :, 0] == out[:, :, 0]).all()          # baseline = R channel of the RGB filter


def numpy_box(img, r):
    """Independent restatement: integer window sums / in-bounds count, truncated."""
    h, w, _ = img.shape
    acc = np.zeros((h, w, 3), np.float32)
    cnt = np.zeros((h, w), np.float32)
    pad = np.zeros((h + 2 * r, w + 2 * r, 3), np.float32)
    pad[r:r + h, r:r + w] = img[:, :, :3]
    one = np.zeros((h + 2 * r, w + 2 * r), np.float32)
    one[r:r + h, r:r + w] = 1
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            acc += pad[dy:dy + h, dx:dx + w]
            cnt += one[dy:dy + h, dx:dx + w]
    return (acc / cnt[:, :, None]).astype(np.uint8)


@pytest.mark.parametrize("shape,r", [((1, 1), 2), ((3, 7), 0), ((5, 4), 3), ((37, 61), 2), ((16, 16), 9)])
def test_edge_shapes_against_numpy(orc, shape, r):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
    out = orc.box_filter(img, r, 1, False)
    assert (out[:, :, :3] == numpy_box(img, r)).all()
    assert (out[:, :, 3] == 0).all()


def test_depth_ping_pong(orc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 24, 4), dtype=np.uint8)
    two = orc.box_filter(img, 1, 2, False)
    once = orc.box_filter(img, 1, 1, False)
    assert (two == orc.box_filter(once, 1, 1, False)).all()
    three = orc.box_filter(img, 1, 3, True)
    ref = orc.box_filter(orc.box_filter(orc.box_filter(img, 1, 1, True), 1, 1, True), 1, 1, True)
    assert (three == ref).all()


def test_threads_do_not_change_result(orc):
    img = orc.load_cornell("render")
    assert (orc.box_filter(img, 2, 1, False, threads=3) == orc.box_filter(img, 2, 1, False)).all()
