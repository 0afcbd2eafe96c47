// svgf_frame.hip — whole-frame SVGF: T + V + n x A with plane routing, and the opaque context
// that owns the cross-frame history planes.  Host code only (launch ordering on one stream).
//
// The pass skeleton follows the reference's level loop (reference src/filter.cu:23-25:
// level 0 reads the input plane, intermediate levels ping-pong between two buffers, the last
// level writes the output) with one kernel launch per level (its single-launch loop races
// across blocks).  History routing is Appendix A.A.4.
//
// Row strips: for final output rows [row0,row1) every earlier pass is run on the rows the later
// passes tap (redundant rows instead of one halo exchange per pass); results are
// bit-identical to a whole-frame run because every pixel sees the same inputs and arithmetic.
#include "common.h"
#include <vector>

namespace rmd {

struct Reach {
    int atrous[16];   // rows above/below [row0,row1) on which iteration i's OUTPUT is needed
    int v, t;         // same for the V and T outputs
    int input;        // current-frame input planes (color, nd, motion)
    int history;      // history planes
};

static int compute_reach(const rmd_svgf_params* p, Reach& r)
{
    if (!p) return fail(RMD_E_NULL, "svgf params is NULL");
    if (p->iterations < 1 || p->iterations > 8) return fail(RMD_E_PARAM, "iterations %d outside [1,8]", p->iterations);
    if (p->hist_iteration < 0 || p->hist_iteration >= p->iterations)
        return fail(RMD_E_PARAM, "hist_iteration %d outside [0,%d)", p->hist_iteration, p->iterations);
    const int n = p->iterations;
    r.atrous[n - 1] = 0;
    for (int i = n - 2; i >= 0; --i) r.atrous[i] = r.atrous[i + 1] + 2 * (1 << (i + 1));
    r.v = r.atrous[0] + 2;                                  // iteration 0 taps +-2 rows
    r.t = r.v + (p->var_radius > 1 ? p->var_radius : 1);    // V taps +-var_radius rows of T's output
    r.input = r.t + 1;                                      // depth gradient reads nd(y+1)
    r.history = r.t + p->max_motion_rows;
    return RMD_OK;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace rmd

using namespace rmd;

struct rmd_svgf_context {
    int width, height, buf_row0, buf_rows;
    size_t plane_bytes;
    float* hist_color[2];
    float* hist_moments[2];
    float* t_color;
    float* v_color;
    float* ping[2];
    unsigned char* tile_flags;
    int cur;            // index of the history set the next frame reads
    bool has_history;
};

extern "C" {

void rmd_svgf_default_params(rmd_svgf_params* p)
{
    if (!p) return;
    p->alpha_color = 0.05f; p->alpha_moments = 0.2f; p->h_max = 32;
    p->k_z = 10.0f; p->k_n = 0.9f; p->max_motion_rows = 64;
    p->var_h_threshold = 4; p->var_radius = 3;
    p->sigma_n = 128.0f; p->sigma_z = 1.0f; p->sigma_l = 4.0f;
    p->iterations = 5; p->hist_iteration = 0; p->atrous_variant = 0;
    p->tv_workgroups = 0; p->atrous_cus = 0;
}

int rmd_svgf_frame_reach(const rmd_svgf_params* p, int reach[4])
{
    if (!reach) return fail(RMD_E_NULL, "rmd_svgf_frame_reach: reach is NULL");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    reach[0] = r.input;
    reach[1] = r.history;
    reach[2] = r.atrous[p->hist_iteration];
    reach[3] = r.t;
    return RMD_OK;
}

static int check_frame_call(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, Reach& r)
{
    if (int e = check_frame_geometry(f)) return e;
    if (int e = compute_reach(p, r)) return e;
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_frame: rows [%d,%d) invalid", row0, row1);
    if (!f->v_color || !f->out_color || !f->hist_color_out) return fail(RMD_E_NULL, "rmd_svgf_frame: v_color/out_color/hist_color_out is NULL");
    const int n = p->iterations;
    const int pingpong_needed = n - 1 - (p->hist_iteration < n - 1 ? 1 : 0);
    if ((pingpong_needed >= 1 && !f->ping[0]) || (pingpong_needed >= 2 && !f->ping[1]))
        return fail(RMD_E_NULL, "rmd_svgf_frame: ping planes are NULL");
    return RMD_OK;
}

int rmd_svgf_frame(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream)
{
    if (int e = rmd_svgf_frame_tv(f, p, row0, row1, stream)) return e;
    return rmd_svgf_frame_atrous(f, p, row0, row1, stream, nullptr);
}

int rmd_svgf_frame_tv(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream)
{
    Reach r;
    if (int e = check_frame_call(f, p, row0, row1, r)) return e;
    const int H = f->height;
    // T also fills v_color, so V only rewrites short-history pixels (when statistics are wanted V
    // runs unfused: it then has to visit every pixel anyway)
    const bool fuse = f->stats == nullptr;
    // one decision for both passes: T skips the t_color stores V will not read
    const bool sparse = variance_reads_sparse_t_color(f, p, fuse);
    const int t0 = clampi(row0 - r.t, 0, H), t1 = clampi(row1 + r.t, 0, H);
    if (int e = launch_temporal(f, p, t0, t1, stream, fuse, sparse)) return e;
    const int v0 = clampi(row0 - r.v, 0, H), v1 = clampi(row1 + r.v, 0, H);
    return launch_variance(f, p, v0, v1, stream, fuse, sparse);
}

int rmd_svgf_frame_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                          void* history_ready_event)
{
    Reach r;
    if (int e = check_frame_call(f, p, row0, row1, r)) return e;
    const int H = f->height;
    const int n = p->iterations;
    const float* in = f->v_color;
    int pp = 0;
    for (int i = 0; i < n; ++i) {
        float* out;
        if (i == n - 1) out = f->out_color;
        else if (i == p->hist_iteration) out = f->hist_color_out;
        else { out = f->ping[pp]; pp ^= 1; }
        const int a0 = clampi(row0 - r.atrous[i], 0, H), a1 = clampi(row1 + r.atrous[i], 0, H);
        if (int e = rmd_svgf_atrous(f, p, i, in, out, a0, a1, stream)) return e;
        if (i == n - 1 && i == p->hist_iteration && f->hist_color_out != out) {
            const size_t off = (size_t)(a0 - f->buf_row0) * f->width * 4;
            RMD_HIP(hipMemcpyAsync(f->hist_color_out + off, out + off, (size_t)(a1 - a0) * f->width * 16,
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
        }
        // next frame's history (hist_color_out, and t_moments since T ran before) is complete here
        if (i == p->hist_iteration && history_ready_event)
            RMD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(history_ready_event), as_stream(stream)));
        in = out;
    }
    return RMD_OK;
}

int rmd_svgf_context_create(int width, int height, int buf_row0, int buf_rows, rmd_svgf_context** out)
{
    if (!out) return fail(RMD_E_NULL, "rmd_svgf_context_create: out is NULL");
    *out = nullptr;
    rmd_svgf_frame_desc g = {};
    g.width = width; g.height = height; g.buf_row0 = buf_row0; g.buf_rows = buf_rows;
    if (int e = check_frame_geometry(&g)) return e;
    rmd_svgf_context* c = new rmd_svgf_context();
    c->width = width; c->height = height; c->buf_row0 = buf_row0; c->buf_rows = buf_rows;
    c->plane_bytes = (size_t)buf_rows * width * 16;
    c->cur = 0; c->has_history = false;
    float** planes[] = { &c->hist_color[0], &c->hist_color[1], &c->hist_moments[0], &c->hist_moments[1],
                         &c->t_color, &c->v_color, &c->ping[0], &c->ping[1] };
    for (float** q : planes) *q = nullptr;
    for (float** q : planes) {
        hipError_t e = hipMalloc((void**)q, c->plane_bytes);
        if (e == hipSuccess) e = hipMemset(*q, 0, c->plane_bytes);
        if (e != hipSuccess) {
            for (float** z : planes) if (*z) (void)hipFree(*z);
            delete c;
            return hip_fail(e, "rmd_svgf_context_create: hipMalloc");
        }
    }
    c->tile_flags = nullptr;
    {
        const size_t fb = RMD_TILE_FLAGS_BYTES(width, height);
        hipError_t e = hipMalloc((void**)&c->tile_flags, fb);
        if (e == hipSuccess) e = hipMemset(c->tile_flags, 0, fb);
        if (e != hipSuccess) {
            for (float** z : planes) if (*z) (void)hipFree(*z);
            if (c->tile_flags) (void)hipFree(c->tile_flags);
            delete c;
            return hip_fail(e, "rmd_svgf_context_create: hipMalloc(tile flags)");
        }
    }
    *out = c;
    return RMD_OK;
}

void rmd_svgf_context_destroy(rmd_svgf_context* c)
{
    if (!c) return;
    if (c->tile_flags) (void)hipFree(c->tile_flags);
    float* planes[] = { c->hist_color[0], c->hist_color[1], c->hist_moments[0], c->hist_moments[1],
                        c->t_color, c->v_color, c->ping[0], c->ping[1] };
    for (float* q : planes) if (q) (void)hipFree(q);
    delete c;
}

int rmd_svgf_context_reset_history(rmd_svgf_context* c, void* stream)
{
    if (!c) return fail(RMD_E_NULL, "rmd_svgf_context_reset_history: ctx is NULL");
    c->has_history = false;
    for (int i = 0; i < 2; ++i) {
        RMD_HIP(hipMemsetAsync(c->hist_color[i], 0, c->plane_bytes, as_stream(stream)));
        RMD_HIP(hipMemsetAsync(c->hist_moments[i], 0, c->plane_bytes, as_stream(stream)));
    }
    return RMD_OK;
}

int rmd_svgf_context_describe(rmd_svgf_context* c, rmd_svgf_frame_desc* f)
{
    if (!c || !f) return fail(RMD_E_NULL, "rmd_svgf_context_describe: NULL argument");
    f->width = c->width; f->height = c->height; f->buf_row0 = c->buf_row0; f->buf_rows = c->buf_rows;
    f->hist_color = c->has_history ? c->hist_color[c->cur] : nullptr;
    f->hist_moments = c->has_history ? c->hist_moments[c->cur] : nullptr;
    f->t_color = c->t_color;
    f->t_moments = c->hist_moments[c->cur ^ 1];
    f->v_color = c->v_color;
    f->hist_color_out = c->hist_color[c->cur ^ 1];
    f->ping[0] = c->ping[0]; f->ping[1] = c->ping[1];
    f->v_tile_flags = c->tile_flags;
    return RMD_OK;
}

int rmd_svgf_context_denoise(rmd_svgf_context* c, const rmd_svgf_params* p, const float* color, const float* nd,
                             const float* motion, const float* prev_nd, float* out, int row0, int row1, void* stream)
{
    if (!c) return fail(RMD_E_NULL, "rmd_svgf_context_denoise: ctx is NULL");
    if (!color || !nd || !motion || !out) return fail(RMD_E_NULL, "rmd_svgf_context_denoise: a required plane is NULL");
    rmd_svgf_frame_desc f = {};
    if (int e = rmd_svgf_context_describe(c, &f)) return e;
    f.color = color; f.nd = nd; f.motion = motion;
    const bool use_hist = c->has_history && prev_nd != nullptr;
    f.prev_nd = use_hist ? prev_nd : nullptr;
    if (!use_hist) { f.hist_color = nullptr; f.hist_moments = nullptr; }
    f.out_color = out;
    f.t_debug = nullptr; f.stats = nullptr;
    if (int e = rmd_svgf_frame(&f, p, row0, row1, stream)) return e;
    c->cur ^= 1;            // this frame's t_moments / hist_color_out become the history
    c->has_history = true;
    return RMD_OK;
}

int rmd_svgf_context_history(rmd_svgf_context* c, float** hist_color, float** hist_moments)
{
    if (!c || !hist_color || !hist_moments) return fail(RMD_E_NULL, "rmd_svgf_context_history: NULL argument");
    *hist_color = c->hist_color[c->cur];
    *hist_moments = c->hist_moments[c->cur];
    return RMD_OK;
}

}  // extern "C"
