// svgf.h — C++ face of the SVGF C ABI (rmd_svgf_* in rmd_api.h).  The reference only names SVGF
// (README.md:3-10); pass semantics are SURVEY.md Appendix A.
#ifndef RMD_SVGF_H
#define RMD_SVGF_H

#include "gbuffer.h"
#include "utils.h"

using SvgfParams = rmd_svgf_params;
using SvgfFrame = rmd_svgf_frame_desc;

inline SvgfParams svgfDefaultParams() { SvgfParams p; rmd_svgf_default_params(&p); return p; }
inline void svgfTemporal(const SvgfFrame& f, const SvgfParams& p, int row0, int row1, void* stream = nullptr) { rmdCheck(rmd_svgf_temporal(&f, &p, row0, row1, stream), "svgfTemporal"); }
inline void svgfVariance(const SvgfFrame& f, const SvgfParams& p, int row0, int row1, void* stream = nullptr) { rmdCheck(rmd_svgf_variance(&f, &p, row0, row1, stream), "svgfVariance"); }
inline void svgfAtrous(const SvgfFrame& f, const SvgfParams& p, int iteration, const float* in, float* out, int row0, int row1, void* stream = nullptr) { rmdCheck(rmd_svgf_atrous(&f, &p, iteration, in, out, row0, row1, stream), "svgfAtrous"); }
inline void svgfFrame(const SvgfFrame& f, const SvgfParams& p, int row0, int row1, void* stream = nullptr) { rmdCheck(rmd_svgf_frame(&f, &p, row0, row1, stream), "svgfFrame"); }

// Owner of the cross-frame history planes (move-only).
class SvgfContext {
    rmd_svgf_context* ctx = nullptr;
public:
    SvgfContext(int width, int height) { rmdCheck(rmd_svgf_context_create(width, height, 0, height, &ctx), "SvgfContext"); }
    SvgfContext(int width, int height, int bufRow0, int bufRows) { rmdCheck(rmd_svgf_context_create(width, height, bufRow0, bufRows, &ctx), "SvgfContext"); }
    SvgfContext(const SvgfContext&) = delete;
    SvgfContext& operator=(const SvgfContext&) = delete;
    SvgfContext(SvgfContext&& o) noexcept : ctx(o.ctx) { o.ctx = nullptr; }
    ~SvgfContext() { rmd_svgf_context_destroy(ctx); }
    void resetHistory(void* stream = nullptr) { rmdCheck(rmd_svgf_context_reset_history(ctx, stream), "resetHistory"); }
    void denoise(const SvgfParams& p, const float* color, const float* nd, const float* motion, const float* prevNd,
                 float* out, int row0, int row1, void* stream = nullptr)
    {
        rmdCheck(rmd_svgf_context_denoise(ctx, &p, color, nd, motion, prevNd, out, row0, row1, stream), "SvgfContext::denoise");
    }
    // The same frame in the parts a mid-frame neighbour exchange cuts it into (rmd_svgf_params.exchange_iteration >= 0):
    // RMD_ATROUS_HEAD = T + V + iterations 0..X, then the caller exchanges midPlane()'s halo rows, RMD_ATROUS_INTERIOR,
    // RMD_ATROUS_TAIL (include/rmd_api.h rmd_svgf_frame_atrous_part; include/strips.h NodeDenoiser drives it).
    void denoisePart(const SvgfParams& p, const float* color, const float* nd, const float* motion, const float* prevNd,
                     float* out, int row0, int row1, void* stream, int part)
    {
        rmdCheck(rmd_svgf_context_denoise_part(ctx, &p, color, nd, motion, prevNd, out, row0, row1, stream, part), "SvgfContext::denoisePart");
    }
    // One frame straight on the reference's frame descriptor (include/gbuffer.h:6-14): uchar4 render / albedo / normal in, uchar4
    // denoised out, six launches; the 8-bit conversions, the demodulation by albedo and the modulation + quantisation run inside
    // the frame's first and last launch (rmd_svgf_gbuffer_frame).  motion: float2 per pixel (current -> previous) or NULL = static.
    void denoise(const GBuffer& frame, const SvgfParams& p, const float* motion = nullptr, float albedoEps = 1.0f / 255.0f, void* stream = nullptr)
    {
        rmdCheck(rmd_svgf_gbuffer_frame(toAbi(frame), ctx, &p, motion, albedoEps, stream), "SvgfContext::denoise(GBuffer)");
    }
    float* midPlane(const SvgfParams& p) { float* q = nullptr; rmdCheck(rmd_svgf_context_mid_plane(ctx, &p, &q), "SvgfContext::midPlane"); return q; }
    rmd_svgf_context* get() { return ctx; }
};

// The reference's intended call shape for its README's goal: SVGF on a GBuffer (README.md:3-10; include/gbuffer.h:6-14).
inline void svgfDenoise(const GBuffer& frame, SvgfContext& ctx, const SvgfParams& p, const float* motion = nullptr,
                        float albedoEps = 1.0f / 255.0f, void* stream = nullptr)
{
    ctx.denoise(frame, p, motion, albedoEps, stream);
}

// hipGraph capture of whatever is queued on `stream` between begin() and end() (rmd_graph_* in rmd_api.h), for a host whose
// own launch path is slow.  Capture an even number of SvgfContext frames (the history planes ping-pong) after at least one
// eager frame, and keep the planes the captured calls named in place.
class FrameGraph {
    void* exec = nullptr;
public:
    FrameGraph() = default;
    FrameGraph(const FrameGraph&) = delete;
    FrameGraph& operator=(const FrameGraph&) = delete;
    ~FrameGraph() { rmd_graph_destroy(exec); }
    // RAII capture: everything queued on `stream` while a Capture lives is recorded; commit() ends the capture and instantiates
    // the graph, and a Capture that goes out of scope without it (an exception between begin and end) ends and DISCARDS the
    // capture instead of leaving the stream in capture mode.
    class Capture {
        FrameGraph& g; void* stream; bool open = true;
    public:
        Capture(FrameGraph& graph, void* s) : g(graph), stream(s) { rmdCheck(rmd_graph_capture_begin(stream), "FrameGraph::Capture"); }
        Capture(const Capture&) = delete;
        Capture& operator=(const Capture&) = delete;
        void commit() { open = false; g.end(stream); }
        ~Capture() { if (open) { void* dropped = nullptr; (void)rmd_graph_capture_end(stream, &dropped); rmd_graph_destroy(dropped); } }
    };
    // The graph holds what the captured calls decided when they were queued: plane pointers, whether the frame had a history (no
    // resetHistory() between capture and replay), the parameters and band plans (the same SvgfParams).
    static void begin(void* stream) { rmdCheck(rmd_graph_capture_begin(stream), "FrameGraph::begin"); }
    void end(void* stream) { rmd_graph_destroy(exec); exec = nullptr; rmdCheck(rmd_graph_capture_end(stream, &exec), "FrameGraph::end"); }
    void launch(void* stream) { rmdCheck(rmd_graph_launch(exec, stream), "FrameGraph::launch"); }
};

#endif
