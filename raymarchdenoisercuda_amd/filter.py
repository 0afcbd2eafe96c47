"""Host-side mirror of the reference's filter interface (include/filter.cuh) over the C ABI.

`filterKernelBaseline` / `filterKernelTiled` keep the reference's names and argument meaning
(a GBuffer and a FilterParams passed by value, reference include/filter.cuh:25-26); launch
geometry is chosen inside the library instead of by the caller (reference src/test.cu:70-71),
and errors raise RmdError instead of being silently dropped (reference src/test.cu:73-89).
torch is used only as the device allocator.
"""
import torch

from ._lib import FilterParams, GBuffer, Int2, check, lib


def _stream_ptr(stream):
    if stream is None:
        return None
    return int(getattr(stream, "cuda_stream", stream))


def filterKernelBaseline(frame: GBuffer, params: FilterParams, stream=None):
    """Replaces `filterKernelBaseline<<<grid,block,smem>>>(frame, params)` (reference src/test.cu:73-75)."""
    check(lib.rmd_filter_baseline(frame, params, _stream_ptr(stream)))


def filterKernelTiled(frame: GBuffer, params: FilterParams, stream=None):
    """Replaces `filterKernelTiled<<<grid,block,smem>>>(frame, params)` (reference src/test.cu:85-87)."""
    check(lib.rmd_filter_tiled(frame, params, _stream_ptr(stream)))


def _check_plane(t, name, shape=None):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.dim() == 3 and t.shape[2] == 4):
        raise ValueError(f"{name}: expected a contiguous CUDA uint8 tensor [H, W, 4]")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)} != {tuple(shape)}")
    return t.data_ptr()


def make_gbuffer(render, denoised, buffer0=None, buffer1=None, normal=None, albedo=None) -> GBuffer:
    """GBuffer over torch uint8 [H, W, 4] CUDA planes (the caller keeps them alive)."""
    h, w, _ = render.shape
    g = GBuffer()
    g.shape = Int2(w, h)
    g.render = _check_plane(render, "render")
    g.denoised = _check_plane(denoised, "denoised", render.shape)
    g.normal = _check_plane(normal, "normal", render.shape)
    g.albedo = _check_plane(albedo, "albedo", render.shape)
    g.buffer[0] = _check_plane(buffer0, "buffer[0]", render.shape)
    g.buffer[1] = _check_plane(buffer1, "buffer[1]", render.shape)
    return g


def box_filter(render: torch.Tensor, params: FilterParams, tiled: bool = True) -> torch.Tensor:
    """Convenience: run the uchar4 filter on a [H, W, 4] uint8 CUDA tensor and return `denoised`."""
    denoised = torch.empty_like(render)
    b0 = torch.empty_like(render) if params.depth > 1 else None
    b1 = torch.empty_like(render) if params.depth > 1 else None
    g = make_gbuffer(render, denoised, b0, b1)
    (filterKernelTiled if tiled else filterKernelBaseline)(g, params)
    return denoised
