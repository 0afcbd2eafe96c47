"""rmd_svgf_gbuffer_frame on the GPU: SVGF in ONE call on the reference's own frame descriptor (include/gbuffer.h:6-14: uchar4
render / albedo / normal in, uchar4 denoised out), its 8-bit ends fused into the first and the last of the frame's six launches.

Two anchors:
  * the unfused chain of eight calls through the same C ABI (3 x rmd_convert_u8_to_f32, rmd_demodulate, the float-plane frame,
    rmd_convert_f32_to_u8): `denoised`, the history planes and T's integer outputs must be IDENTICAL, byte for byte;
  * the oracle's composition of the same steps (tests/oracle_lib.py gbuffer_frame): integer outputs bit-exact, float planes
    within the frame tolerance 5e-4 (1 + |ref|), bytes within 1 LSB (a float inside the tolerance can sit on a rounding boundary).
PARITY UNPINNED BY THE REFERENCE for SVGF itself: the reference has the struct, not the filter.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 1.0 / 255.0
TOL_FRAME = 5e-4


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def tiled_cornell_sequence(orc, width, height, frames, pan, seed=7):
    """uint8 planes of `frames` frames: the Cornell fixture planes tiled to the frame, rolled by `pan` pixels per frame (the
    tiled picture is periodic, so a roll is a camera pan), the render multiplied by per-frame re-seeded noise."""
    planes = {n: orc.load_cornell(n) for n in ("render", "albedo", "normal")}
    reps = (-(-height // 500), -(-width // 500), 1)
    tiled = {n: np.tile(a, reps)[:height, :width].copy() for n, a in planes.items()}
    rng = np.random.default_rng(seed)
    seq = []
    for f in range(frames):
        r, a, n = (np.roll(tiled[k], (f * pan[1], f * pan[0]), axis=(0, 1)) for k in ("render", "albedo", "normal"))
        noise = 0.75 + 0.5 * rng.random((height, width, 1), dtype=np.float32)
        r = r.copy()
        r[..., :3] = np.clip(r[..., :3].astype(np.float32) * noise, 0, 255).astype(np.uint8)
        seq.append((np.ascontiguousarray(r), np.ascontiguousarray(a), np.ascontiguousarray(n)))
    return seq


def run_chain(rmd, seq, motion, p, width, height):
    """The eight-call chain, frame by frame; returns (bytes per frame, the SvgfDenoiser with debug planes)."""
    den = rmd.SvgfDenoiser(width, height, params=p, debug=True)
    outs, dbg = [], []
    m = motion if motion is not None else torch.zeros((height, width, 2), dtype=torch.float32, device="cuda")
    keep = []
    for render, albedo, normal in seq:
        color = rmd.svgf.convert_u8_to_f32(dev(render), False, 0.0)
        alb = rmd.svgf.convert_u8_to_f32(dev(albedo), False, 0.0)
        nd = rmd.svgf.convert_u8_to_f32(dev(normal), True, -1.0)
        rmd.svgf.demodulate(color, alb, EPS, out=color)
        out = den.denoise(color, nd, m)
        outs.append(rmd.svgf.convert_f32_to_u8(out, alb))
        dbg.append(den.t_debug.clone())
        keep.append(nd)                       # nd is borrowed as prev_nd until the next call
    torch.cuda.synchronize()
    return outs, dbg, den


def run_fused(rmd, seq, motion, p, width, height):
    den = rmd.GBufferDenoiser(width, height, params=p, albedo_eps=EPS, debug=True)
    outs, dbg = [], []
    for render, albedo, normal in seq:
        outs.append(den.frame(dev(render), dev(albedo), dev(normal), motion=motion))
        torch.cuda.synchronize()
        dbg.append(den.t_debug.clone())
    return outs, dbg, den


def test_u8_to_unit_float_on_the_device_is_the_ieee_quotient(rmd, cuda):
    """All 256 bytes in every channel through rmd_convert_u8_to_f32 (csrc/pixel_convert.h unit_from_u8, the division-free
    form the fused front end shares with it): the bits of numpy's float32 division."""
    ramp = np.arange(256, dtype=np.uint8)
    img = np.stack([ramp, ramp[::-1], np.roll(ramp, 37), np.roll(ramp, 101)], axis=-1).reshape(16, 16, 4)
    got = rmd.svgf.convert_u8_to_f32(dev(img), False, -1.0).cpu().numpy()
    want = img.astype(np.float32) / np.float32(255.0)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_quantisation_special_values(rmd, orc, cuda):
    """rmd_convert_f32_to_u8 (csrc/pixel_convert.h: v_floor_f32 + v_cvt_pk_u8_f32, shared with the last a-trous launch of
    rmd_svgf_gbuffer_frame) against the oracle's C cast chain on everything the clamp has to catch: NaN, +-inf, negatives, values
    above 1, and every k/255 +- 1 ulp (the rounding boundaries of x * 255 + 0.5)."""
    k = np.arange(256, dtype=np.float32) / np.float32(255.0)
    edge = np.concatenate([k, np.nextafter(k, np.float32(2)), np.nextafter(k, np.float32(-1)),
                           (np.arange(256, dtype=np.float32) + np.float32(0.5)) / np.float32(255.0)])
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, -1e-30, 1e-30, -5.0, 1.0, 1.0000001, 7.0, 3e38, -3e38, 0.00196, 0.99804], np.float32)
    vals = np.concatenate([edge, special]).astype(np.float32)
    n = (len(vals) + 2) // 3 * 3
    vals = np.resize(vals, n)
    img = np.zeros((1, n, 4), np.float32)
    img[0, :, 0], img[0, :, 1], img[0, :, 2] = vals, np.roll(vals, 1), np.roll(vals, 2)
    with np.errstate(invalid="ignore"):
        want = orc.convert_f32_to_u8(img)
    got = rmd.svgf.convert_f32_to_u8(dev(img)).cpu().numpy()
    assert (got == want).all(), np.argwhere(got != want)[:5]
    alb = np.full((1, n, 4), 0.5, np.float32)
    assert (rmd.svgf.convert_f32_to_u8(dev(img), dev(alb)).cpu().numpy() == orc.convert_f32_to_u8(img, alb)).all()


@pytest.mark.parametrize("width,height,frames,pan,motion_xy", [
    (500, 500, 3, (0, 0), None),                 # the fixture itself, static camera, motion = NULL
    (500, 500, 3, (0, 0), (0.0, 0.0)),           # the same with an explicit zero motion plane
    (300, 203, 3, (2, 1), (-2.25, -1.5)),        # ragged size, fractional reprojection
    (1920, 1080, 3, (2, 1), (-2.25, -0.75)),     # 1080p tiled, fractional pan
    (3840, 2160, 3, (3, 1), (-2.75, -1.25)),     # 4K tiled, fractional pan
])
def test_gbuffer_frame_is_the_eight_call_chain_byte_for_byte(rmd, orc, cuda, width, height, frames, pan, motion_xy):
    p = rmd.default_params()
    p.max_motion_rows = 8
    seq = tiled_cornell_sequence(orc, width, height, frames, pan)
    motion = None
    if motion_xy is not None:
        motion = torch.empty((height, width, 2), dtype=torch.float32, device="cuda")
        motion[..., 0], motion[..., 1] = motion_xy
    chain, chain_dbg, den_c = run_chain(rmd, seq, motion, p, width, height)
    fused, fused_dbg, den_f = run_fused(rmd, seq, motion, p, width, height)
    for f in range(frames):
        assert torch.equal(chain_dbg[f], fused_dbg[f]), f"frame {f}: T's integer outputs differ on {(chain_dbg[f] != fused_dbg[f]).any(-1).sum().item()} pixels"
        assert torch.equal(chain[f], fused[f]), f"frame {f}: {(chain[f] != fused[f]).sum().item()} of {chain[f].numel()} bytes differ"
    # the cross-frame state after the last frame
    for a, b, name in zip(den_c.history(), den_f.history(), ("hist_color", "hist_moments", "hist_len")):
        assert torch.equal(a, b), f"{name}: {(a != b).sum().item()} values differ"
    if frames > 1 and motion_xy not in (None, (0.0, 0.0)):
        h = fused_dbg[-1][..., 3]
        assert (h > 1).float().mean().item() > 0.3, "the sequence must exercise the history path"
        assert (fused_dbg[-1][..., 2] != 15).any().item(), "and its rejections"


def test_gbuffer_frame_against_the_oracle_composition(rmd, orc, cuda):
    """Cornell 500x500, three static frames: convert -> demodulate -> orc_svgf_frame -> modulate + quantise on the CPU."""
    p = rmd.default_params()
    render, albedo, normal = (orc.load_cornell(n) for n in ("render", "albedo", "normal"))
    rng = np.random.default_rng(3)
    den = rmd.GBufferDenoiser(500, 500, params=p, albedo_eps=EPS, debug=True)
    hist = None
    for f in range(3):
        r = render.copy()
        r[..., :3] = np.clip(r[..., :3].astype(np.float32) * (0.75 + 0.5 * rng.random((500, 500, 1), dtype=np.float32)), 0, 255).astype(np.uint8)
        want, fr = orc.gbuffer_frame(r, albedo, normal, p, hist=hist, albedo_eps=EPS, threads=8)
        got = den.frame(dev(r), dev(albedo), dev(normal)).cpu().numpy()
        assert (den.t_debug.cpu().numpy() == fr.t_debug).all(), f"frame {f}: reprojection index / tap mask / history length"
        hc, hm, hl = (t.cpu().numpy() for t in den.history())
        assert (hm == fr.t_moments).all() and (hl == fr.t_len).all(), f"frame {f}: T's float outputs and the history length are bit-exact"
        err = np.abs(hc.astype(np.float64) - fr.hist_color_out) / (1.0 + np.abs(fr.hist_color_out))
        assert err.max() <= TOL_FRAME, f"frame {f}: hist_color {err.max():.2e}"
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1, f"frame {f}: a byte differs by {diff.max()}"
        assert (diff != 0).mean() < 0.005, f"frame {f}: {(diff != 0).mean():.4f} of the bytes differ"
        assert (got[..., 3] == 255).all()
        hist = fr.history()


@pytest.mark.parametrize("iterations,hist_iteration", [(5, 4), (1, 0), (3, 1), (2, 0), (6, 0)])
def test_gbuffer_frame_other_iteration_counts(rmd, orc, cuda, iterations, hist_iteration):
    """The byte-storing last launch at every step (1, 2, 4 ... and the direct kernel beyond iteration 4), and the case where
    the last iteration is also the history iteration (floats to the history plane + a conversion launch)."""
    width, height = 260, 190
    p = rmd.default_params()
    p.iterations, p.hist_iteration, p.max_motion_rows = iterations, hist_iteration, 8
    seq = tiled_cornell_sequence(orc, width, height, 2, (1, 1))
    motion = torch.empty((height, width, 2), dtype=torch.float32, device="cuda")
    motion[..., 0], motion[..., 1] = -1.25, -1.0
    chain, chain_dbg, den_c = run_chain(rmd, seq, motion, p, width, height)
    fused, fused_dbg, den_f = run_fused(rmd, seq, motion, p, width, height)
    for f in range(2):
        assert torch.equal(chain_dbg[f], fused_dbg[f])
        assert torch.equal(chain[f], fused[f]), f"frame {f}: {(chain[f] != fused[f]).sum().item()} bytes differ"
    for a, b in zip(den_c.history(), den_f.history()):
        assert torch.equal(a, b)


def test_gbuffer_frame_direct_kernel_and_history_reset(rmd, orc, cuda):
    width, height = 200, 120
    p = rmd.default_params()
    p.atrous_variant = 1                         # the direct kernel stores the bytes
    seq = tiled_cornell_sequence(orc, width, height, 2, (0, 0))
    chain, _, _ = run_chain(rmd, seq, None, p, width, height)
    fused, _, den = run_fused(rmd, seq, None, p, width, height)
    assert torch.equal(chain[0], fused[0]) and torch.equal(chain[1], fused[1])
    # after a reset the next frame is a first frame again
    den.reset_history()
    again = den.frame(dev(seq[0][0]), dev(seq[0][1]), dev(seq[0][2]))
    torch.cuda.synchronize()
    assert torch.equal(again, fused[0])
    assert (den.t_debug[..., 3] == 1).all()


def test_gbuffer_frame_refusals(rmd, orc, cuda):
    width, height = 128, 64
    seq = tiled_cornell_sequence(orc, width, height, 1, (0, 0))
    r, a, n = (dev(x) for x in seq[0])
    den = rmd.GBufferDenoiser(width, height)
    with pytest.raises(ValueError):
        den.frame(r[:32], a[:32], n[:32])                                   # another shape than the context's
    g = rmd.make_gbuffer(r[:32].contiguous(), torch.empty_like(r[:32]), normal=n[:32].contiguous(), albedo=a[:32].contiguous())
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(den.params), None, EPS, None) == -2          # RMD_E_SHAPE
    g = rmd.make_gbuffer(r, torch.empty_like(r), normal=n)                  # no albedo
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(den.params), None, EPS, None) == -1          # RMD_E_NULL
    g = rmd.make_gbuffer(r, r, normal=n, albedo=a)                          # denoised aliases render
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(den.params), None, EPS, None) == -4          # RMD_E_BUFFER
    g = rmd.make_gbuffer(r, torch.empty_like(r), normal=n, albedo=a)
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(den.params), None, 0.0, None) == -3          # eps must be > 0
    p = rmd.default_params()
    p.exchange_iteration = 3
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(p), None, EPS, None) == -3                   # whole frames only
    p = rmd.default_params()
    p.var_radius = 2
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, den._ctx, C.byref(p), None, EPS, None) == -6                   # RMD_E_UNSUPPORTED
    # a context that holds a row strip cannot take a GBuffer
    ctx = C.c_void_p()
    rmd.check(rmd.lib.rmd_svgf_context_create(width, height, 16, 32, C.byref(ctx)))
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, ctx, C.byref(den.params), None, EPS, None) == -5               # RMD_E_ROWS
    rmd.lib.rmd_svgf_context_destroy(ctx)
    # and nothing above has disturbed the context
    den.frame(r, a, n)
    torch.cuda.synchronize()


def test_gbuffer_and_float_frames_alternate_on_one_context(rmd, orc, cuda):
    """include/rmd_api.h: frames of rmd_svgf_gbuffer_frame and of rmd_svgf_context_denoise may alternate on one context; the first
    frame after a switch starts a new history (the previous frame's nd is a plane the other path cannot name), later ones accumulate."""
    width, height = 192, 128
    p = rmd.default_params()
    seq = tiled_cornell_sequence(orc, width, height, 1, (0, 0))
    r, a, n = (dev(x) for x in seq[0])
    den = rmd.GBufferDenoiser(width, height, params=p, debug=True)
    color = rmd.svgf.convert_u8_to_f32(r, False, 0.0)
    nd = rmd.svgf.convert_u8_to_f32(n, True, -1.0)
    motion = torch.zeros((height, width, 2), dtype=torch.float32, device="cuda")
    out = torch.empty_like(color)

    def float_frame(prev_nd):
        rmd.check(rmd.lib.rmd_svgf_context_denoise(den._ctx, C.byref(p), color.data_ptr(), nd.data_ptr(), motion.data_ptr(),
                                                   None if prev_nd is None else prev_nd.data_ptr(), out.data_ptr(), 0, height, None))
        torch.cuda.synchronize()
        return int(den.t_debug[..., 3].min().item()), int(den.t_debug[..., 3].max().item())

    def u8_frame():
        den.frame(r, a, n)
        torch.cuda.synchronize()
        return int(den.t_debug[..., 3].min().item()), int(den.t_debug[..., 3].max().item())

    assert u8_frame() == (1, 1)
    assert u8_frame() == (2, 2)                  # static camera: every pixel reprojects onto itself
    assert float_frame(nd) == (1, 1)             # switch: a new history although a prev_nd was handed in
    assert float_frame(nd) == (2, 2)
    assert float_frame(nd) == (3, 3)
    assert u8_frame() == (1, 1)                  # and back
    assert u8_frame() == (2, 2)
    den.reset_history()
    assert u8_frame() == (1, 1)


def test_gbuffer_frames_replayed_from_a_graph_equal_eager_frames(rmd, orc, cuda):
    """The one-call path under rmd_graph_*: two frames (the context's history and nd planes ping-pong) captured on a side stream
    behind two eager ones (the first call allocates the context's nd planes) and replayed give the eager sequence's bytes."""
    width, height = 320, 200
    p = rmd.default_params()
    seq = tiled_cornell_sequence(orc, width, height, 2, (1, 0))
    fa, fb = (tuple(dev(x) for x in fr) for fr in seq)
    motion = torch.zeros((height, width, 2), dtype=torch.float32, device="cuda")
    motion[..., 0] = -1.0
    eager = rmd.GBufferDenoiser(width, height, params=p)
    want = torch.empty_like(fa[0])
    for _ in range(4):                                   # A B | A B A B A B
        eager.frame(*fa, want, motion)
        eager.frame(*fb, want, motion)
    torch.cuda.synchronize()
    den = rmd.GBufferDenoiser(width, height, params=p)
    got = torch.zeros_like(fa[0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        den.frame(*fa, got, motion)
        den.frame(*fb, got, motion)
        side.synchronize()
    with rmd.capture(side) as g:
        den.frame(*fa, got, motion)
        den.frame(*fb, got, motion)
    side.synchronize()
    assert not torch.equal(got, want)                    # the captured frames have not run
    for _ in range(3):
        g.launch()
    side.synchronize()
    assert torch.equal(got, want), f"{(got != want).sum().item()} bytes differ"
    g.destroy()


def test_context_denoise_part_all_is_context_denoise(rmd, cuda):
    """rmd_svgf_context_denoise_part(RMD_ATROUS_ALL) is documented as 'everything in order': T + V included.  (It used to skip
    them and rotate the history over a stale frame.)  Three frames through both entry points, bit for bit."""
    width, height, frames = 192, 130, 3
    p = rmd.default_params()
    p.max_motion_rows = 8
    inputs = [rmd.svgf.synth_gbuffer(width, height, f) for f in range(frames)]
    outs = {}
    for form in ("denoise", "part_all"):
        ctx = C.c_void_p()
        rmd.check(rmd.lib.rmd_svgf_context_create(width, height, 0, height, C.byref(ctx)))
        res = []
        for f, (c, nd, m) in enumerate(inputs):
            out = torch.empty_like(c)
            prev = inputs[f - 1][1].data_ptr() if f else None
            if form == "denoise":
                rmd.check(rmd.lib.rmd_svgf_context_denoise(ctx, C.byref(p), c.data_ptr(), nd.data_ptr(), m.data_ptr(), prev, out.data_ptr(), 0, height, None))
            else:
                rmd.check(rmd.lib.rmd_svgf_context_denoise_part(ctx, C.byref(p), c.data_ptr(), nd.data_ptr(), m.data_ptr(), prev, out.data_ptr(), 0, height, None,
                                                                 rmd.svgf.ATROUS_ALL))
            torch.cuda.synchronize()
            res.append(out)
        rmd.lib.rmd_svgf_context_destroy(ctx)
        outs[form] = res
    for f in range(frames):
        assert torch.equal(outs["denoise"][f], outs["part_all"][f]), f"frame {f}"
    assert not torch.equal(outs["denoise"][2], outs["denoise"][0])


def test_a_strip_with_a_mid_frame_exchange_cannot_run_unexchanged(rmd, cuda):
    """exchange_iteration >= 0 on a strip without the exchange would read halo rows nobody delivered: RMD_E_PARAM from the C ABI,
    ValueError from SvgfDenoiser.denoise without hooks; whole frames are unaffected."""
    width, height = 128, 400
    p = rmd.default_params()
    p.max_motion_rows = 8
    p.exchange_iteration = 3
    c, nd, m = rmd.svgf.synth_gbuffer(width, height, 0)
    den = rmd.SvgfDenoiser(width, height, params=p)
    den.denoise(c, nd, m)                                                   # whole frame: fine
    with pytest.raises(ValueError):
        den.denoise(c, nd, m, row0=100, row1=300)
    d = den.describe(c, nd, m, torch.empty_like(c))
    assert rmd.lib.rmd_svgf_frame(C.byref(d), C.byref(p), 100, 300, None) == -3
    assert rmd.lib.rmd_svgf_frame_atrous(C.byref(d), C.byref(p), 100, 300, None, None) == -3
    assert rmd.lib.rmd_svgf_frame(C.byref(d), C.byref(p), 0, height, None) == 0
    torch.cuda.synchronize()
