"""CPU checks of the one-call GBuffer path (rmd_svgf_gbuffer_frame): the arithmetic its fused 8-bit front end substitutes
for a division, and the argument checks that run before anything touches HIP."""
import ctypes as C

import numpy as np


def test_division_free_u8_to_unit_float_equals_the_ieee_quotient(orc):
    """csrc/pixel_convert.h unit_from_u8: q = v * RN(1/255); e = fma(-q, 255, v); q + e * RN(1/255) -- restated with C99 fmaf in
    oracle/svgf_oracle.c and compared with (float)b / 255.0f for all 256 bytes."""
    assert orc.lib.orc_unit_from_u8_mismatches() == 0
    # the plain product WITHOUT the correction step is NOT the quotient (which is why the correction is there)
    v = np.arange(256, dtype=np.float32)
    assert (v * np.float32(1.0 / 255.0) != v / np.float32(255.0)).sum() > 0


def test_oracle_composition_of_the_gbuffer_frame(orc):
    """orc.gbuffer_frame = convert -> demodulate -> SVGF -> modulate + quantise, on the Cornell planes: the picture survives
    (mean preserved), the noise does not, alpha is opaque, and a second static frame accumulates history."""
    render, albedo, normal = (orc.load_cornell(n)[100:228, 150:278].copy() for n in ("render", "albedo", "normal"))
    p = orc.default_params()
    out0, fr0 = orc.gbuffer_frame(render, albedo, normal, p)
    out1, fr1 = orc.gbuffer_frame(render, albedo, normal, p, hist=fr0.history())
    assert out0.dtype == np.uint8 and (out0[..., 3] == 255).all()
    assert abs(out0[..., :3].mean() - render[..., :3].mean()) < 4.0
    rough = lambda a: np.abs(np.diff(a[..., 0].astype(np.int32), axis=1)).mean()      # noqa: E731
    assert rough(out0) < 0.6 * rough(render)
    assert (fr0.t_debug[..., 3] == 1).all() and (fr1.t_debug[..., 3] == 2).all()       # static camera: every pixel reprojects onto itself
    assert (fr0.nd[..., 3] == 1.0).all()                                               # opaque alpha = depth 1


def test_gbuffer_frame_argument_checks(rmd):
    """Everything rmd_svgf_gbuffer_frame can refuse before it touches the device (no GPU needed: the context is never created)."""
    p = rmd.default_params()
    g = rmd.GBuffer()
    assert rmd.lib.rmd_svgf_gbuffer_frame(g, None, C.byref(p), None, 1.0 / 255.0, None) == -1          # RMD_E_NULL: no context
    assert b"ctx" in rmd.lib.rmd_last_error_string()
    assert rmd.lib.rmd_svgf_context_set_debug_plane(None, None) == -1
