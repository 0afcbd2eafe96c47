"""Host side of formulations that were measured and lost; they need the experiments build of the
library (`make experiments`, RMD_LIB_PATH=build/variants/librmd_experiments.so; the product library answers RMD_E_UNSUPPORTED).
Kept out of the product module `svgf.py`."""
import ctypes as C

import torch

from ._lib import check, lib
from .filter import _stream_ptr
from .svgf import SvgfDenoiser, frame_mid_exchange


class NextFrameDenoiser(SvgfDenoiser):
    """The temporal pass of frame k+1 as a side job of frame k's a-trous launches (rmd_svgf_frame_atrous_next): pass the
    FOLLOWING call's (color, nd, motion) as `next_frame`; that call then skips T + V.  Serial form, no hooks."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._tv_done_for = None     # data pointers of the frame whose T + V the previous call already ran
        self._keep_next = None

    def reset_history(self):
        super().reset_history()
        self._tv_done_for = None

    def denoise(self, color, nd, motion, out=None, row0=None, row1=None, stream=None, next_frame=None):
        if out is None:
            out = torch.empty_like(color)
        tv_done = self._tv_done_for is not None
        if tv_done and self._tv_done_for != tuple(t.data_ptr() for t in (color, nd, motion)):
            raise ValueError("the previous denoise() call ran this frame's temporal pass on the planes it was given as next_frame; "
                             "call denoise() with those planes")
        self._tv_done_for = None
        if next_frame is None:
            return super().denoise(color, nd, motion, out, row0, row1, stream, _tv_done=tv_done)
        if self.pipelined or frame_mid_exchange(self.params)[0] >= 0:
            raise ValueError("next_frame needs the serial frame without a mid-frame exchange")
        row0 = max(self.buf_row0, 0) if row0 is None else row0
        row1 = min(self.buf_row0 + self.buf_rows, self.height) if row1 is None else row1
        d = self.describe(color, nd, motion, out)
        s_ptr = _stream_ptr(torch.cuda.current_stream() if stream is None else stream)
        p = self.params
        if not tv_done:
            check(lib.rmd_svgf_frame_tv(C.byref(d), C.byref(p), row0, row1, s_ptr))
        nc, nnd, nm = next_frame
        dn = self.describe(nc, nnd, nm, out, ahead=nd)
        check(lib.rmd_svgf_frame_atrous_next(C.byref(d), C.byref(p), row0, row1, s_ptr, None, C.byref(dn)))
        self._tv_done_for = tuple(t.data_ptr() for t in (nc, nnd, nm))
        self._keep_next = next_frame                 # the planes stay referenced until the following call
        self.cur ^= 1
        self.has_history = True
        self.prev_nd = nd
        return out
