"""ctypes binding of librmd.so — the C ABI declared in include/rmd_api.h.

The library is the product: there is NO CPU fallback.  Importing this module without a built
`lib/librmd.so` raises ImportError telling how to build it (`make lib` / `__graft_entry__.build()`).
"""
import ctypes as C
import os

# torch ships its own HIP runtime (libamdhip64.so under torch/lib).  It has to be in the process
# before librmd.so is dlopen()ed, otherwise librmd pulls in /opt/rocm's copy first and the two
# runtimes disagree about the devices ("no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401  (device allocator for the host mirror; imported here for load order)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RMD_LIB_PATH selects an alternative build of the same library (kernel tuning experiments only)
LIB_PATH = os.environ.get("RMD_LIB_PATH") or os.path.join(_HERE, "lib", "librmd.so")


class RmdError(RuntimeError):
    """Non-zero return of a librmd entry point (the C++ wrappers throw std::runtime_error,
    mirroring what the reference harness catches at src/test.cu:40-42)."""

    def __init__(self, code, message):
        super().__init__(f"librmd error {code}: {message}")
        self.code = code


# ---- PODs (layouts static_asserted on the C++ side, include/filter.h, include/gbuffer.h) ----
class Int2(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int)]


class GBuffer(C.Structure):
    """reference include/gbuffer.h:6-14 (56 bytes)."""
    _fields_ = [("shape", Int2), ("render", C.c_void_p), ("denoised", C.c_void_p), ("normal", C.c_void_p),
                ("albedo", C.c_void_p), ("buffer", C.c_void_p * 2)]


class FilterParams(C.Structure):
    """reference include/filter.cuh:11-23 (36 bytes); cacheInput/cacheBuffer default true."""
    AVERAGE, GAUSSIAN, CROSS, WAVELET = 0, 1, 2, 3
    _fields_ = [("type", C.c_int), ("depth", C.c_int), ("level", C.c_int), ("radius", C.c_int),
                ("sigmaSpace", C.c_float), ("sigmaColor", C.c_float), ("sigmaAlbedo", C.c_float),
                ("sigmaNormal", C.c_float), ("cacheInput", C.c_ubyte), ("cacheBuffer", C.c_ubyte)]

    def __init__(self, type=0, depth=1, level=0, radius=2, sigmaSpace=0.0, sigmaColor=0.0, sigmaAlbedo=0.0,
                 sigmaNormal=0.0, cacheInput=True, cacheBuffer=True):
        super().__init__(type, depth, level, radius, sigmaSpace, sigmaColor, sigmaAlbedo, sigmaNormal,
                         int(bool(cacheInput)), int(bool(cacheBuffer)))


class SvgfParams(C.Structure):
    _fields_ = [("alpha_color", C.c_float), ("alpha_moments", C.c_float), ("h_max", C.c_int),
                ("k_z", C.c_float), ("k_n", C.c_float), ("max_motion_rows", C.c_int),
                ("var_h_threshold", C.c_int), ("var_radius", C.c_int),
                ("sigma_n", C.c_float), ("sigma_z", C.c_float), ("sigma_l", C.c_float),
                ("iterations", C.c_int), ("hist_iteration", C.c_int), ("atrous_variant", C.c_int),
                ("tv_workgroups", C.c_int), ("atrous_cus", C.c_int), ("exchange_iteration", C.c_int)]


class SvgfFrameDesc(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buf_row0", C.c_int), ("buf_rows", C.c_int),
                ("color", C.c_void_p), ("nd", C.c_void_p), ("motion", C.c_void_p),
                ("hist_color", C.c_void_p), ("hist_moments", C.c_void_p), ("hist_len", C.c_void_p), ("prev_nd", C.c_void_p),
                ("t_color", C.c_void_p), ("t_moments", C.c_void_p), ("t_len", C.c_void_p), ("t_debug", C.c_void_p),
                ("v_color", C.c_void_p), ("hist_color_out", C.c_void_p), ("ping", C.c_void_p * 2),
                ("out_color", C.c_void_p), ("stats", C.c_void_p), ("v_tile_flags", C.c_void_p)]


class StripPlan(C.Structure):
    """include/rmd_api.h rmd_strip_plan (mirrors sharding.StripPlan)."""
    _fields_ = [("height", C.c_int), ("world", C.c_int), ("rank", C.c_int), ("row0", C.c_int), ("row1", C.c_int),
                ("buf_row0", C.c_int), ("buf_rows", C.c_int), ("reach_in", C.c_int), ("reach_hist", C.c_int),
                ("have_color", C.c_int), ("have_moments", C.c_int), ("mid_iteration", C.c_int), ("mid_rows", C.c_int)]


class HaloStep(C.Structure):
    RECV, SEND = 0, 1
    PLANE_HIST_COLOR, PLANE_HIST_MOMENTS, PLANE_MID, PLANE_HIST_LEN = 0, 1, 2, 3
    _fields_ = [("kind", C.c_int), ("plane", C.c_int), ("row_lo", C.c_int), ("row_hi", C.c_int), ("peer", C.c_int)]


class SynthDesc(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("buf_row0", C.c_int), ("buf_rows", C.c_int),
                ("seed", C.c_uint32), ("frame", C.c_int), ("pan_x", C.c_float), ("pan_y", C.c_float)]


# name -> (restype, argtypes).  Every symbol include/rmd_api.h declares is listed here and
# tests/test_abi.py checks the library exports each of them.
_P = C.c_void_p
SYMBOLS = {
    "rmd_filter_baseline": (C.c_int, [GBuffer, FilterParams, _P]),
    "rmd_filter_tiled": (C.c_int, [GBuffer, FilterParams, _P]),
    "rmd_svgf_default_params": (None, [C.POINTER(SvgfParams)]),
    "rmd_svgf_temporal": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P]),
    "rmd_svgf_variance": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P]),
    "rmd_svgf_atrous": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "rmd_svgf_atrous2": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rmd_debug_atrous_protocol_errors": (C.c_int, [C.POINTER(C.c_uint)]),
    "rmd_debug_atrous_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rmd_svgf_frame": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P]),
    "rmd_svgf_frame_tv": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P]),
    "rmd_svgf_frame_atrous": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P, _P]),
    "rmd_svgf_frame_atrous_next": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P, _P,
                                             C.POINTER(SvgfFrameDesc)]),
    "rmd_svgf_frame_atrous_part": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.c_int, _P, _P, C.c_int]),
    "rmd_svgf_frame_mid_exchange": (C.c_int, [C.POINTER(SvgfParams), C.POINTER(C.c_int * 2)]),
    "rmd_svgf_frame_iteration_reach": (C.c_int, [C.POINTER(SvgfParams), C.POINTER(C.c_int * 8)]),
    "rmd_svgf_frame_iteration_plane": (C.c_int, [C.POINTER(SvgfFrameDesc), C.POINTER(SvgfParams), C.c_int, C.POINTER(_P)]),
    "rmd_svgf_frame_reach": (C.c_int, [C.POINTER(SvgfParams), C.POINTER(C.c_int * 4)]),
    "rmd_svgf_context_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "rmd_svgf_context_destroy": (None, [_P]),
    "rmd_svgf_context_reset_history": (C.c_int, [_P, _P]),
    "rmd_svgf_context_denoise": (C.c_int, [_P, C.POINTER(SvgfParams), _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "rmd_svgf_context_denoise_part": (C.c_int, [_P, C.POINTER(SvgfParams), _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "rmd_svgf_context_mid_plane": (C.c_int, [_P, C.POINTER(SvgfParams), C.POINTER(_P)]),
    "rmd_svgf_gbuffer_frame": (C.c_int, [GBuffer, _P, C.POINTER(SvgfParams), _P, C.c_float, _P]),
    "rmd_svgf_context_set_debug_plane": (C.c_int, [_P, _P]),
    "rmd_svgf_context_history": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "rmd_svgf_context_describe": (C.c_int, [_P, C.POINTER(SvgfFrameDesc)]),
    "rmd_strip_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rmd_strip_plan_make": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(SvgfParams), C.POINTER(StripPlan)]),
    "rmd_halo_plan": (C.c_int, [C.POINTER(StripPlan), C.POINTER(HaloStep), C.c_int, C.POINTER(C.c_int)]),
    "rmd_halo_bytes": (C.c_size_t, [C.POINTER(StripPlan), C.c_int]),
    "rmd_mid_halo_plan": (C.c_int, [C.POINTER(StripPlan), C.POINTER(HaloStep), C.c_int, C.POINTER(C.c_int)]),
    "rmd_mid_exchange": (C.c_int, [_P, C.POINTER(StripPlan), C.c_int, _P, _P]),
    "rmd_mid_exchange_all": (C.c_int, [_P, C.POINTER(StripPlan), C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "rmd_comm_available": (C.c_int, []),
    "rmd_comm_unique_id": (C.c_int, [_P]),
    "rmd_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "rmd_comm_create_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_P)]),
    "rmd_comm_destroy": (C.c_int, [_P]),
    "rmd_halo_exchange": (C.c_int, [_P, C.POINTER(StripPlan), C.c_int, _P, _P, _P, _P]),
    "rmd_halo_exchange_all": (C.c_int, [_P, C.POINTER(StripPlan), C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "rmd_halo_exchange_steps": (C.c_int, [_P, C.c_int, C.POINTER(HaloStep), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "rmd_exchange_steps": (C.c_int, [_P, C.c_int, C.POINTER(HaloStep), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P), _P]),
    "rmd_convert_u8_to_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_float, _P]),
    "rmd_convert_f32_to_u8": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "rmd_demodulate": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_float, _P]),
    "rmd_synth_gbuffer": (C.c_int, [C.POINTER(SynthDesc), _P, _P, _P, _P, _P]),
    "rmd_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "rmd_free": (C.c_int, [_P]),
    "rmd_memset": (C.c_int, [_P, C.c_int, C.c_size_t, _P]),
    "rmd_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t]),
    "rmd_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t]),
    "rmd_memcpy_d2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "rmd_memcpy_h2d_async": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "rmd_memcpy_d2h_async": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "rmd_host_alloc_pinned": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "rmd_host_free_pinned": (C.c_int, [_P]),
    "rmd_stream_create": (C.c_int, [C.POINTER(_P)]),
    "rmd_stream_destroy": (C.c_int, [_P]),
    "rmd_stream_sync": (C.c_int, [_P]),
    "rmd_event_create": (C.c_int, [C.POINTER(_P)]),
    "rmd_event_destroy": (C.c_int, [_P]),
    "rmd_event_record": (C.c_int, [_P, _P]),
    "rmd_event_synchronize": (C.c_int, [_P]),
    "rmd_stream_wait_event": (C.c_int, [_P, _P]),
    "rmd_graph_capture_begin": (C.c_int, [_P]),
    "rmd_graph_capture_end": (C.c_int, [_P, C.POINTER(_P)]),
    "rmd_graph_launch": (C.c_int, [_P, _P]),
    "rmd_graph_destroy": (C.c_int, [_P]),
    "rmd_device_sync": (C.c_int, []),
    "rmd_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rmd_set_device": (C.c_int, [C.c_int]),
    "rmd_print_device_properties": (C.c_int, []),
    "rmd_last_error_string": (C.c_char_p, []),
    "rmd_version": (C.c_char_p, []),
    "rmd_has_experiments": (C.c_int, []),
    "rmd_timer_create": (C.c_int, [C.POINTER(_P)]),
    "rmd_timer_destroy": (C.c_int, [_P]),
    "rmd_timer_start": (C.c_int, [_P, _P]),
    "rmd_timer_stop": (C.c_int, [_P, _P]),
    "rmd_timer_elapsed_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product and has no fallback. "
            "Build it with `make lib` (hipcc --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of sync
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def last_error():
    return lib.rmd_last_error_string().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise RmdError(rc, last_error())
