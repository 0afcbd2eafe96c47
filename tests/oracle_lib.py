"""numpy/ctypes front end of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.  Builds the oracle with `make -C oracle` if the shared object is missing.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")



def build_oracle():
    """`make` is a no-op when liboracle.so is newer than its sources; without gcc/make (never the
    case in this image) an existing .so is used as is."""
    try:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(ORACLE_SO):
            raise
    return ORACLE_SO


def structs_from_layout(text):
    """ctypes structures built from oracle/abi_probe.c's report of include/rmd_api.h as the C compiler lays it out
    (offsetof / sizeof per field) -- independently of the product's Python binding, so that a wrong field order in
    raymarchdenoisercuda_amd/_lib.py cannot be wrong identically on the oracle's side (tests/test_abi.py compares the two)."""
    scalar = {"i": C.c_int, "u": C.c_uint32, "f": C.c_float, "b": C.c_ubyte, "p": C.c_void_p}
    out, name, size, fields = {}, None, 0, []

    def close():
        if name is None:
            return
        ct, pos, pad = [], 0, 0
        for fname, off, fsize, kind in fields:
            if off > pos:
                ct.append((f"_pad{pad}", C.c_ubyte * (off - pos)))
                pad += 1
            base = out[kind[2:]] if kind.startswith("s:") else scalar[kind]
            n = fsize // C.sizeof(base)
            assert n * C.sizeof(base) == fsize, (name, fname)
            ct.append((fname, base if n == 1 else base * n))
            pos = off + fsize
        cls = type(name, (C.Structure,), {"_fields_": ct})
        assert C.sizeof(cls) == size, (name, C.sizeof(cls), size)
        for fname, off, fsize, _ in fields:
            assert getattr(cls, fname).offset == off and getattr(cls, fname).size == fsize, (name, fname)
        out[name] = cls

    for line in text.splitlines():
        if not line.strip():
            continue
        parts = line.split()
        if line[0] != " ":
            close()
            name, size, fields = parts[0], int(parts[1]), []
        else:
            fields.append((parts[0], int(parts[1]), int(parts[2]), parts[3]))
    close()
    return out


def _load():
    build_oracle()
    lib = C.CDLL(ORACLE_SO)
    lib.orc_abi_layout.restype = C.c_char_p
    global ABI_LAYOUT, STRUCTS, SvgfFrameDesc, SvgfParams, SynthDesc, FilterParamsC
    ABI_LAYOUT = lib.orc_abi_layout().decode()
    STRUCTS = structs_from_layout(ABI_LAYOUT)
    SvgfFrameDesc, SvgfParams, SynthDesc = STRUCTS["rmd_svgf_frame_desc"], STRUCTS["rmd_svgf_params"], STRUCTS["rmd_synth_desc"]
    FilterParamsC = STRUCTS["rmd_filter_params"]
    P = C.c_void_p
    lib.orc_box_level.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_box_filter.argtypes = [P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_box_filter_mt.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_svgf_temporal.argtypes = [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int]
    lib.orc_svgf_variance.argtypes = [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int]
    lib.orc_svgf_atrous.argtypes = [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, P, P, C.c_int, C.c_int]
    lib.orc_svgf_frame.argtypes = [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int]
    lib.orc_svgf_pass_mt.argtypes = [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, P, P, C.c_int]
    lib.orc_synth_gbuffer.argtypes = [C.POINTER(SynthDesc), P, P, P, P]
    lib.orc_hash32.argtypes = [C.c_uint32] * 4
    lib.orc_hash32.restype = C.c_uint32
    lib.orc_convert_u8_to_f32.argtypes = [P, P, C.c_size_t, C.c_int, C.c_float]
    lib.orc_convert_f32_to_u8.argtypes = [P, P, P, C.c_size_t]
    lib.orc_demodulate.argtypes = [P, P, P, C.c_size_t, C.c_float]
    lib.orc_demodulate.restype = None
    lib.orc_hardware_threads.restype = C.c_int
    lib.orc_unit_from_u8_mismatches.restype = C.c_int
    for f in (lib.orc_box_level, lib.orc_box_filter, lib.orc_box_filter_mt, lib.orc_svgf_temporal, lib.orc_svgf_variance,
              lib.orc_svgf_atrous, lib.orc_svgf_frame, lib.orc_svgf_pass_mt, lib.orc_synth_gbuffer,
              lib.orc_convert_u8_to_f32, lib.orc_convert_f32_to_u8):
        f.restype = None
    return lib


lib = _load()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _by_name(cls, obj):
    """`obj` as the oracle's own structure `cls`.  A structure of the product binding (raymarchdenoisercuda_amd._lib)
    is copied FIELD BY FIELD BY NAME, never as bytes: if the binding's layout were wrong, the oracle would still get
    the values the test set by name and the parity test would fail instead of agreeing on the wrong meaning."""
    if obj is None or isinstance(obj, cls):
        return obj
    out = cls()
    names = [n for n, _ in cls._fields_ if not n.startswith("_pad")]
    assert names == [n for n, _ in obj._fields_], (cls.__name__, "field names differ from the binding's")
    for n in names:
        v = getattr(obj, n)
        if hasattr(v, "__len__"):
            for k in range(len(v)):
                getattr(out, n)[k] = v[k]
        else:
            setattr(out, n, v)
    return out


def default_params():
    """SURVEY Appendix A defaults (kept independent of librmd's rmd_svgf_default_params; a test compares them)."""
    return SvgfParams(alpha_color=0.05, alpha_moments=0.2, h_max=32, k_z=10.0, k_n=0.9, max_motion_rows=64,
                      var_h_threshold=4, var_radius=3, sigma_n=128.0, sigma_z=1.0, sigma_l=4.0,
                      iterations=5, hist_iteration=0, atrous_variant=0, tv_workgroups=0, atrous_cus=0, exchange_iteration=-1)


# ---- box filter ------------------------------------------------------------------------------
def box_filter(render_rgba: np.ndarray, radius=2, depth=1, gray_from_r=False, threads=1) -> np.ndarray:
    """render_rgba: uint8 [H, W, 4].  gray_from_r=True = filterKernelBaseline, False = filterKernelTiled."""
    assert render_rgba.dtype == np.uint8 and render_rgba.ndim == 3 and render_rgba.shape[2] == 4
    src = np.ascontiguousarray(render_rgba)
    h, w, _ = src.shape
    out = np.zeros_like(src)
    if depth == 1 and threads > 1:
        lib.orc_box_filter_mt(_p(src), _p(out), w, h, radius, int(gray_from_r), threads)
        return out
    b0 = np.zeros_like(src) if depth > 1 else None
    b1 = np.zeros_like(src) if depth > 1 else None
    lib.orc_box_filter(_p(src), _p(out), _p(b0), _p(b1), w, h, radius, depth, int(gray_from_r))
    return out


def weighted_filter(render_rgba: np.ndarray, params, normal=None, albedo=None) -> np.ndarray:
    """FilterParams::type GAUSSIAN / CROSS / WAVELET (`params` = the ctypes FilterParams)."""
    src = np.ascontiguousarray(render_rgba)
    h, w, _ = src.shape
    out, b0, b1 = np.zeros_like(src), np.zeros_like(src), np.zeros_like(src)
    nrm = None if normal is None else np.ascontiguousarray(normal)
    alb = None if albedo is None else np.ascontiguousarray(albedo)
    lib.orc_weighted_filter.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
    lib.orc_weighted_filter.restype = None
    lib.orc_weighted_filter(_p(src), _p(out), _p(b0), _p(b1), _p(nrm), _p(alb), w, h, C.byref(_by_name(FilterParamsC, params)))
    return out


# ---- SVGF ------------------------------------------------------------------------------------
class Frame:
    """Host planes of one SVGF frame (numpy float32) + the descriptor the oracle reads."""

    def __init__(self, width, height, color, nd, motion, hist_color=None, hist_moments=None, prev_nd=None, debug=True, hist_len=None):
        """History of the previous frame: (hist_color, hist_moments, prev_nd) + hist_len -- or hist_moments given as the
        PAIR (t_moments, t_len) of the previous Frame (see `history()`)."""
        if isinstance(hist_moments, tuple):
            hist_moments, hist_len = hist_moments
        self.width, self.height = width, height
        f4 = lambda: np.zeros((height, width, 4), np.float32)  # noqa: E731
        self.color = np.ascontiguousarray(color, np.float32)
        self.nd = np.ascontiguousarray(nd, np.float32)
        self.motion = np.ascontiguousarray(motion, np.float32)
        self.hist_color = None if hist_color is None else np.ascontiguousarray(hist_color, np.float32)
        self.hist_moments = None if hist_moments is None else np.ascontiguousarray(hist_moments, np.float32)
        self.hist_len = None if hist_len is None else np.ascontiguousarray(hist_len, np.uint8)
        assert (self.hist_moments is None) == (self.hist_len is None), "hist_moments and hist_len come together"
        assert self.hist_moments is None or (self.hist_moments.shape == (height, width, 2) and self.hist_len.shape == (height, width))
        self.prev_nd = None if prev_nd is None else np.ascontiguousarray(prev_nd, np.float32)
        self.t_color, self.v_color = f4(), f4()
        self.t_moments = np.zeros((height, width, 2), np.float32)        # float2 (m1, m2)
        self.t_len = np.zeros((height, width), np.uint8)                 # history length
        self.t_debug = np.zeros((height, width, 4), np.int32) if debug else None
        self.hist_color_out, self.out_color = f4(), f4()
        self.ping = [f4(), f4()]
        self.desc = SvgfFrameDesc()
        d = self.desc
        d.width, d.height, d.buf_row0, d.buf_rows = width, height, 0, height
        for name in ("color", "nd", "motion", "hist_color", "hist_moments", "hist_len", "prev_nd", "t_color", "t_moments", "t_len",
                     "t_debug", "v_color", "hist_color_out", "out_color"):
            arr = getattr(self, name)
            setattr(d, name, None if arr is None else arr.ctypes.data)
        d.ping[0], d.ping[1] = self.ping[0].ctypes.data, self.ping[1].ctypes.data

    def history(self):
        """(hist_color, (hist_moments, hist_len), prev_nd) for the next frame's Frame(...)."""
        return self.hist_color_out, (self.t_moments, self.t_len), self.nd


def temporal(fr: Frame, p, row0=0, row1=None):
    lib.orc_svgf_temporal(C.byref(fr.desc), C.byref(_by_name(SvgfParams, p)), row0, fr.height if row1 is None else row1)


def variance(fr: Frame, p, row0=0, row1=None):
    lib.orc_svgf_variance(C.byref(fr.desc), C.byref(_by_name(SvgfParams, p)), row0, fr.height if row1 is None else row1)


def atrous(fr: Frame, p, iteration, src: np.ndarray, dst: np.ndarray, row0=0, row1=None):
    assert src.dtype == np.float32 and dst.dtype == np.float32 and src.flags.c_contiguous and dst.flags.c_contiguous
    lib.orc_svgf_atrous(C.byref(fr.desc), C.byref(_by_name(SvgfParams, p)), iteration, _p(src), _p(dst), row0, fr.height if row1 is None else row1)


def frame(fr: Frame, p, threads=1):
    lib.orc_svgf_frame(C.byref(fr.desc), C.byref(_by_name(SvgfParams, p)), threads)


def pass_mt(fr: Frame, p, which, iteration=0, src=None, dst=None, threads=1):
    lib.orc_svgf_pass_mt(C.byref(fr.desc), C.byref(_by_name(SvgfParams, p)), which, iteration, _p(src), _p(dst), threads)


def synth_gbuffer(width, height, frame_index, buf_row0=0, buf_rows=None, seed=1234, pan=(1.25, -0.5), want_albedo=False):
    buf_rows = height if buf_rows is None else buf_rows
    color = np.zeros((buf_rows, width, 4), np.float32)
    nd = np.zeros((buf_rows, width, 4), np.float32)
    motion = np.zeros((buf_rows, width, 2), np.float32)
    albedo = np.zeros((buf_rows, width, 4), np.float32) if want_albedo else None
    d = SynthDesc(width, height, buf_row0, buf_rows, seed, frame_index, pan[0], pan[1])
    lib.orc_synth_gbuffer(C.byref(d), _p(color), _p(nd), _p(motion), _p(albedo))
    return (color, nd, motion, albedo) if want_albedo else (color, nd, motion)


def convert_u8_to_f32(src_u8, renormalize_xyz=False, w_value=-1.0):
    src = np.ascontiguousarray(src_u8)
    out = np.zeros(src.shape, np.float32)
    lib.orc_convert_u8_to_f32(_p(src), _p(out), src.shape[0] * src.shape[1], int(renormalize_xyz), w_value)
    return out


def convert_f32_to_u8(src_f32, albedo=None):
    src = np.ascontiguousarray(src_f32, np.float32)
    alb = None if albedo is None else np.ascontiguousarray(albedo, np.float32)
    out = np.zeros(src.shape, np.uint8)
    lib.orc_convert_f32_to_u8(_p(src), _p(alb), _p(out), src.shape[0] * src.shape[1])
    return out


def demodulate(radiance, albedo, eps=1e-3):
    r = np.ascontiguousarray(radiance, np.float32)
    a = np.ascontiguousarray(albedo, np.float32)
    out = np.zeros(r.shape, np.float32)
    lib.orc_demodulate(_p(r), _p(a), _p(out), r.shape[0] * r.shape[1], eps)
    return out


def gbuffer_frame(render_u8, albedo_u8, normal_u8, p, motion=None, hist=None, albedo_eps=1.0 / 255.0, threads=4):
    """The composition rmd_svgf_gbuffer_frame is defined as (include/rmd_api.h): orc_convert_u8_to_f32 (render, albedo: c/255;
    normal: renormalised, w/255 = depth) -> orc_demodulate -> orc_svgf_frame -> orc_convert_f32_to_u8 (x albedo).
    hist = (hist_color, hist_moments, prev_nd) of the previous frame or None.  Returns (denoised uint8, Frame)."""
    h, w, _ = render_u8.shape
    color = demodulate(convert_u8_to_f32(render_u8, False, 0.0), convert_u8_to_f32(albedo_u8, False, 0.0), albedo_eps)
    albedo = convert_u8_to_f32(albedo_u8, False, 0.0)
    nd = convert_u8_to_f32(normal_u8, True, -1.0)
    m = np.zeros((h, w, 2), np.float32) if motion is None else motion
    hc, hm, pn = hist if hist is not None else (None, None, None)
    fr = Frame(w, h, color, nd, m, hc, hm, pn)
    frame(fr, p, threads=threads)
    return convert_f32_to_u8(fr.out_color, albedo), fr


def hardware_threads():
    return int(lib.orc_hardware_threads())


# ---- fixtures ---------------------------------------------------------------------------------
def load_cornell(name="render"):
    """tests/golden/cornell/<name>.png -> uint8 [500, 500, 4] RGBA with A=255 (what the reference's
    Image(path, 4) yields, reference src/image.cpp:33-40)."""
    from PIL import Image
    im = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "cornell", f"{name}.png")).convert("RGB"))
    return np.concatenate([im, np.full(im.shape[:2] + (1,), 255, np.uint8)], axis=2).copy()


def cornell_svgf_inputs():
    """Cornell planes as SVGF inputs: color = render/255 (a=0), nd.xyz = renormalised
    normal/255 (zero stays zero), nd.w = 1 (depth.png is saturated), motion = 0."""
    color = convert_u8_to_f32(load_cornell("render"), False, 0.0)
    nd = convert_u8_to_f32(load_cornell("normal"), True, 1.0)
    motion = np.zeros(color.shape[:2] + (2,), np.float32)
    return color, nd, motion
