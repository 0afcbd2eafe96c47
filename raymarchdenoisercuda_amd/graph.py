"""hipGraph capture of queued launches (rmd_graph_* in include/rmd_api.h), for hosts whose launch path is the bottleneck
(this package's frame loop is not: tools/graph_probe.py measures replay == eager within run-to-run noise).

    with rmd.capture(stream) as g:          # stream: a torch.cuda.Stream (not the default stream), current inside the block
        den.denoise(*frame_a, out=out)      # nothing runs here; an EVEN number of frames (the history planes ping-pong)
        den.denoise(*frame_b, out=out)
    g.launch()                              # on `stream`; the planes the captured calls named must still be alive and in place

The first eager frame on a device sets kernel attributes and must have run before a capture.  A graph bakes in what the captured
calls decided when they were queued -- the plane POINTERS, whether the frame had a history (so: no reset_history() between
capture and replay, and capture behind at least one eager frame), the parameters and the band plans (so: the same
rmd_svgf_params, atrous_cus included)."""
import ctypes as C
import sys

import torch

from ._lib import check, lib


class Graph:
    """An instantiated hipGraph (hipGraphExec_t) behind the C ABI."""

    def __init__(self, handle, stream):
        self._h, self.stream = handle, stream

    def launch(self, stream=None):
        s = self.stream if stream is None else stream
        check(lib.rmd_graph_launch(self._h, C.c_void_p(s.cuda_stream)))

    def destroy(self):
        if self._h is not None:
            check(lib.rmd_graph_destroy(self._h))
            self._h = None

    def __del__(self):
        if sys.is_finalizing():            # the HIP runtime may already be gone: a crash in there cannot be caught
            return
        try:
            self.destroy()
        except Exception:
            pass


class capture:
    """Context manager: everything queued on `stream` inside the block is captured instead of executed."""

    def __init__(self, stream):
        if stream is None or stream.cuda_stream == 0:
            raise ValueError("the default stream cannot be captured: pass a torch.cuda.Stream()")
        self.stream, self.graph = stream, None
        self._ctx = torch.cuda.stream(stream)

    def __enter__(self):
        self._ctx.__enter__()
        check(lib.rmd_graph_capture_begin(C.c_void_p(self.stream.cuda_stream)))
        self.graph = Graph(None, self.stream)
        return self.graph

    def __exit__(self, exc_type, exc, tb):
        h = C.c_void_p()
        rc = lib.rmd_graph_capture_end(C.c_void_p(self.stream.cuda_stream), C.byref(h))
        self._ctx.__exit__(exc_type, exc, tb)
        if exc_type is None:
            check(rc)
            self.graph._h = h
        elif h:
            lib.rmd_graph_destroy(h)
        return False
