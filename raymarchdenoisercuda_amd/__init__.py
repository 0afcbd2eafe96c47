"""raymarchdenoisercuda_amd — MI355X (gfx950) native filter / SVGF hot path.

Drop-in for the hot path of VictorHerbert/RaymarchDenoiserCuda (include/filter.cuh entry points,
gbuffer.h / image.h host structs) plus the SVGF passes its README names.  All compute is in the
hand-written HIP kernels of lib/librmd.so behind the C ABI of include/rmd_api.h; this package is
the Python host-side mirror used by tests/ and bench.py (the C++ mirror is include/*.h +
raymarchdenoisercuda_amd/host/).  There is no CPU fallback: importing fails if the library is
not built.
"""
from ._lib import (FilterParams, GBuffer, Int2, LIB_PATH, RmdError, SvgfFrameDesc, SvgfParams, SynthDesc,
                   check, last_error, lib)
from .filter import box_filter, filterKernelBaseline, filterKernelTiled, make_gbuffer
from . import sharding, svgf
from .svgf import GBufferDenoiser, SvgfDenoiser, default_params
from .graph import Graph, capture

__all__ = ["FilterParams", "GBuffer", "Int2", "LIB_PATH", "RmdError", "SvgfFrameDesc", "SvgfParams", "SynthDesc",
           "check", "last_error", "lib", "box_filter", "filterKernelBaseline", "filterKernelTiled", "make_gbuffer",
           "sharding", "svgf", "GBufferDenoiser", "SvgfDenoiser", "default_params", "Graph", "capture", "HAS_EXPERIMENTS", "ATROUS_VARIANTS"]


# a-trous formulations this build of librmd.so can run (include/rmd_api.h rmd_svgf_params.atrous_variant): the product
# library has the default (0 = 3) and its direct cross-check (1); `make experiments` adds the ones that lost
HAS_EXPERIMENTS = bool(lib.rmd_has_experiments())
ATROUS_VARIANTS = (0, 1, 3) + ((2, 4, 5, 6, 7, 8) if HAS_EXPERIMENTS else ())


def version():
    return lib.rmd_version().decode()
