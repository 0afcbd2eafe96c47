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
#include "svgf_tv.h"
#include <vector>

namespace rmd {

struct Reach {
    int atrous[16];   // rows above/below [row0,row1) on which this call COMPUTES iteration i's output
    int need[16];     // rows above/below [row0,row1) on which iteration i's output is NEEDED by the later passes
    int v, t;         // same for the V and T outputs
    int input;        // current-frame input planes (color, nd, motion)
    int history;      // history planes
    int mid;          // p->exchange_iteration, or -1: the iteration whose output is completed by a neighbour exchange
    int mid_rows;     // need[mid]: rows per side that travel
};

// Redundant rows instead of per-pass halo exchanges: iteration i's output is needed on
// need[i] = need[i+1] + 2*2^(i+1) rows beyond the strip, and a strip computes exactly that.  With ONE neighbour
// exchange inside the frame (rmd_svgf_params.exchange_iteration = X, SURVEY §8e "Halo sizes"), iteration X is computed
// on the strip's own rows only, its need[X] halo rows arrive from rank +-1, and everything in front of X shrinks with it.
static int compute_reach(const rmd_svgf_params* p, Reach& r)
{
    if (!p) return fail(RMD_E_NULL, "svgf params is NULL");
    if (p->iterations < 1 || p->iterations > 8) return fail(RMD_E_PARAM, "iterations %d outside [1,8]", p->iterations);
    if (p->hist_iteration < 0 || p->hist_iteration >= p->iterations)
        return fail(RMD_E_PARAM, "hist_iteration %d outside [0,%d)", p->hist_iteration, p->iterations);
    const int n = p->iterations;
    if (p->exchange_iteration < -1 || p->exchange_iteration > n - 2)
        return fail(RMD_E_PARAM, "exchange_iteration %d outside [-1,%d] (the last iteration has nothing behind it to exchange for)",
                    p->exchange_iteration, n - 2);
    r.mid = p->exchange_iteration;
    r.need[n - 1] = 0;
    for (int i = n - 2; i >= 0; --i) r.need[i] = r.need[i + 1] + 2 * (1 << (i + 1));
    for (int i = n - 1; i >= 0; --i) {
        if (i > r.mid || r.mid < 0) r.atrous[i] = r.need[i];
        else if (i == r.mid)        r.atrous[i] = 0;
        else                        r.atrous[i] = r.atrous[i + 1] + 2 * (1 << (i + 1));
    }
    r.mid_rows = r.mid >= 0 ? r.need[r.mid] : 0;
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
    size_t plane_bytes;         // of a float4 plane
    float* hist_color[2];
    float* hist_moments[2];     // float2 (m1, m2): plane_bytes / 2
    unsigned char* hist_len[2]; // uint8 history length: plane_bytes / 16
    float* t_color;
    float* v_color;
    float* ping[2];
    unsigned char* tile_flags;
    int cur;            // index of the history set the next frame reads
    bool has_history;
    // rmd_svgf_gbuffer_frame: the float (normal, depth) planes its front end writes, this frame's and the previous one's
    // (allocated by the first such call), and whether the previous frame went through that call (nd[nd_cur ^ 1] is its nd)
    float* nd[2];
    int nd_cur;
    bool nd_valid;
    int* t_debug;       // optional int4 plane (rmd_svgf_context_set_debug_plane): T's bit-exact outputs of every frame
};

// A strip (rows beyond [row0,row1) exist in the frame) with a mid-frame exchange cannot run its a-trous iterations in one go:
// the iteration behind the exchanged one would read halo rows nobody delivered.
static int refuse_unexchanged_strip(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, const char* who)
{
    if (p && p->exchange_iteration >= 0 && (row0 > 0 || row1 < f->height))
        return fail(RMD_E_PARAM, "%s: rows [%d,%d) are a strip of a %d-row frame and exchange_iteration = %d: drive the frame in parts "
                    "(RMD_ATROUS_HEAD, the neighbour exchange, RMD_ATROUS_INTERIOR, RMD_ATROUS_TAIL)", who, row0, row1, f->height, p->exchange_iteration);
    return RMD_OK;
}

extern "C" {

void rmd_svgf_default_params(rmd_svgf_params* p)
{
    if (!p) return;
    p->alpha_color = 0.05f; p->alpha_moments = 0.2f; p->h_max = 32;
    p->k_z = 10.0f; p->k_n = 0.9f; p->max_motion_rows = 64;
    p->var_h_threshold = 4; p->var_radius = 3;
    p->sigma_n = 128.0f; p->sigma_z = 1.0f; p->sigma_l = 4.0f;
    p->iterations = 5; p->hist_iteration = 0; p->atrous_variant = 0;
    p->tv_workgroups = 0; p->atrous_cus = 0; p->exchange_iteration = -1;
}

int rmd_svgf_frame_reach(const rmd_svgf_params* p, int reach[4])
{
    if (!reach) return fail(RMD_E_NULL, "rmd_svgf_frame_reach: reach is NULL");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    reach[0] = r.input;
    reach[1] = r.history;
    // hist_color_out rows a rank holds after the frame: what it computed, or -- when the history iteration is the
    // exchanged one -- its own rows plus the halo it received
    reach[2] = p->hist_iteration == r.mid ? r.mid_rows : r.atrous[p->hist_iteration];
    reach[3] = r.t;
    return RMD_OK;
}

int rmd_svgf_frame_iteration_reach(const rmd_svgf_params* p, int reach[8])
{
    if (!reach) return fail(RMD_E_NULL, "rmd_svgf_frame_iteration_reach: reach is NULL");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    for (int i = 0; i < 8; ++i) reach[i] = i < p->iterations ? r.atrous[i] : 0;
    return RMD_OK;
}

int rmd_svgf_frame_mid_exchange(const rmd_svgf_params* p, int mid[2])
{
    if (!mid) return fail(RMD_E_NULL, "rmd_svgf_frame_mid_exchange: mid is NULL");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    mid[0] = r.mid;
    mid[1] = r.mid_rows;
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
    if (f && f->height > 0)
        if (int e = refuse_unexchanged_strip(f, p, row0, row1, "rmd_svgf_frame")) return e;      // before anything is launched
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
    const int t0 = clampi(row0 - r.t, 0, H), t1 = clampi(row1 + r.t, 0, H);
    const int v0 = clampi(row0 - r.v, 0, H), v1 = clampi(row1 + r.v, 0, H);
    // no statistics, the default window, one workgroup per tile: T and V in ONE launch (svgf_temporal_variance_kernel)
    if (fuse && p->var_radius == 3 && p->tv_workgroups == 0 && p->var_h_threshold <= 256 && tuning_env("RMD_FUSED_TV", 1))
        return launch_temporal_variance(f, p, t0, t1, v0, v1, stream);
    // one decision for both passes: T skips the t_color stores V will not read
    const bool sparse = variance_reads_sparse_t_color(f, p, fuse);
    if (int e = launch_temporal(f, p, t0, t1, stream, fuse, sparse)) return e;
    return launch_variance(f, p, v0, v1, stream, fuse, sparse);
}

// plane routing of the a-trous iterations (Appendix A.A.4): in / out plane of iteration i
static void atrous_route(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int i, const float** in, float** out)
{
    const int n = p->iterations;
    const float* src = f->v_color;
    int pp = 0;
    for (int k = 0; k <= i; ++k) {
        float* dst;
        if (k == n - 1) dst = f->out_color;
        else if (k == p->hist_iteration) dst = f->hist_color_out;
        else { dst = f->ping[pp]; pp ^= 1; }
        if (k == i) { *in = src; *out = dst; }
        src = dst;
    }
}

int rmd_svgf_frame_iteration_plane(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration, float** plane)
{
    if (!f || !plane) return fail(RMD_E_NULL, "rmd_svgf_frame_iteration_plane: NULL argument");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    if (iteration < 0 || iteration >= p->iterations) return fail(RMD_E_PARAM, "rmd_svgf_frame_iteration_plane: iteration %d", iteration);
    const float* in;
    atrous_route(f, p, iteration, &in, plane);
    return RMD_OK;
}

// EXPERIMENTS BUILD (measured and lost).
// The a-trous iterations of frame `f` with the temporal pass of the NEXT frame as their side job (svgf_atrous.hip): the
// iterations behind hist_iteration (whose output T(next) reads as history) carry one 64x4 tile of T(next) per workgroup every
// few steps; what they leave over runs as a launch of its own, then V(next).  Afterwards `next` is where rmd_svgf_frame_tv
// would have left it: the caller goes on with rmd_svgf_frame_atrous[_next](next, ...).
int rmd_svgf_frame_atrous_next(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                               void* history_ready_event, const rmd_svgf_frame_desc* next)
{
#ifndef RMD_EXPERIMENTS
    (void)f; (void)p; (void)row0; (void)row1; (void)stream; (void)history_ready_event; (void)next;
    return fail(RMD_E_UNSUPPORTED, "rmd_svgf_frame_atrous_next is an experiment that lost (DESIGN.md section 4.7): make experiments; rmd_has_experiments() == 0 here");
#else
    Reach r;
    if (int e = check_frame_call(f, p, row0, row1, r)) return e;
    if (!next) return fail(RMD_E_NULL, "rmd_svgf_frame_atrous_next: next is NULL");
    if (int e = check_frame_call(next, p, row0, row1, r)) return e;
    if (r.mid >= 0) return fail(RMD_E_PARAM, "rmd_svgf_frame_atrous_next: not with a mid-frame exchange (exchange_iteration >= 0)");
    if (next->stats) return fail(RMD_E_PARAM, "rmd_svgf_frame_atrous_next: statistics frames run T and V on their own (rmd_svgf_frame_tv)");
    if (p->tv_workgroups != 0) return fail(RMD_E_PARAM, "rmd_svgf_frame_atrous_next: tv_workgroups must be 0");
    if (next->width != f->width || next->height != f->height || next->buf_row0 != f->buf_row0 || next->buf_rows != f->buf_rows)
        return fail(RMD_E_SHAPE, "rmd_svgf_frame_atrous_next: the two frames differ in geometry");
    // what T(next) and V(next) write must be none of the planes the carrying iterations still read or write
    const void* busy[] = { f->ping[0], f->ping[1], f->out_color, f->hist_color_out, f->nd };
    const void* written[] = { next->t_color, next->t_moments, next->v_color, next->t_debug };
    for (const void* w : written)
        for (const void* b : busy)
            if (w && w == b) return fail(RMD_E_BUFFER, "rmd_svgf_frame_atrous_next: a plane T / V of the next frame write is still in use by this frame's iterations");
    const int H = f->height, n = p->iterations;
    const int t0 = clampi(row0 - r.t, 0, H), t1 = clampi(row1 + r.t, 0, H);
    const int v0 = clampi(row0 - r.v, 0, H), v1 = clampi(row1 + r.v, 0, H);
    const bool sparse = variance_reads_sparse_t_color(next, p, true);
    AtrousSide side;
    if (int e = make_temporal_args(next, p, t0, t1, true, sparse, &side.t)) return e;
    side.units = side.t.tiles_x * ((t1 - 1) / 4 - t0 / 4 + 1);
    side.counter = side_counter_on_device();
    if (!side.counter) return hip_fail(hipErrorOutOfMemory, "rmd_svgf_frame_atrous_next: device counter");
    RMD_HIP(hipMemsetAsync(side.counter, 0, sizeof(unsigned), as_stream(stream)));
    // one tile per workgroup every `every` steps: the carrying launches' steps (512 pixels each) over the tiles
    double steps = 0.0;
    for (int i = p->hist_iteration + 1; i < n && i <= 4; ++i)
        steps += (double)(clampi(row1 + r.atrous[i], 0, H) - clampi(row0 - r.atrous[i], 0, H)) * f->width / 512.0;
    side.every = steps >= side.units ? (int)(steps / side.units) : 1;
    for (int i = 0; i < n; ++i) {
        const float* in;
        float* out;
        atrous_route(f, p, i, &in, &out);
        const int a0 = clampi(row0 - r.atrous[i], 0, H), a1 = clampi(row1 + r.atrous[i], 0, H);
        const bool carries = i > p->hist_iteration && i <= 4 && p->atrous_variant == 0;
        if (int e = launch_atrous(f, p, i, in, out, a0, a1, 0, 0, stream, carries ? &side : nullptr)) return e;
        if (i == n - 1 && i == p->hist_iteration && f->hist_color_out != out) {
            const size_t off = (size_t)(a0 - f->buf_row0) * f->width * 4;
            RMD_HIP(hipMemcpyAsync(f->hist_color_out + off, out + off, (size_t)(a1 - a0) * f->width * 16,
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
        }
        if (i == p->hist_iteration && history_ready_event)
            RMD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(history_ready_event), as_stream(stream)));
    }
    if (int e = launch_temporal_claim(side, stream)) return e;
    return launch_variance(next, p, v0, v1, stream, true, sparse);
#endif
}

int rmd_svgf_frame_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                          void* history_ready_event)
{
    return rmd_svgf_frame_atrous_part(f, p, row0, row1, stream, history_ready_event, RMD_ATROUS_ALL);
}

int rmd_svgf_frame_atrous_part(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream,
                               void* history_ready_event, int part)
{
    Reach r;
    if (int e = check_frame_call(f, p, row0, row1, r)) return e;
    if (part < RMD_ATROUS_ALL || part > RMD_ATROUS_TAIL) return fail(RMD_E_PARAM, "rmd_svgf_frame_atrous_part: part %d", part);
    if (part != RMD_ATROUS_ALL && r.mid < 0)
        return fail(RMD_E_PARAM, "rmd_svgf_frame_atrous_part: parts need rmd_svgf_params.exchange_iteration >= 0");
    if (part == RMD_ATROUS_ALL)
        if (int e = refuse_unexchanged_strip(f, p, row0, row1, "rmd_svgf_frame_atrous")) return e;
    const int H = f->height;
    const int n = p->iterations;
    for (int i = 0; i < n; ++i) {
        const float* in;
        float* out;
        atrous_route(f, p, i, &in, &out);
        const int a0 = clampi(row0 - r.atrous[i], 0, H), a1 = clampi(row1 + r.atrous[i], 0, H);
        // The exchanged iteration (mid) is computed on the strip's own rows [a0,a1) = [row0,row1), BOUNDARY FIRST: the mid_rows
        // rows at each end that has a neighbour are what that neighbour is waiting for; they go out while the interior rows
        // are computed.  (Cutting the iteration BEHIND the exchange into interior + boundary instead was measured: at step 16
        // the two 32-row launches cost 30 us and the 476-row interior as much as the whole strip)
        int lo = a0, hi = a1;                       // interior of iteration mid
        if (r.mid >= 0 && i == r.mid) {
            if (row0 > 0) lo = clampi(row0 + r.mid_rows, a0, a1);
            if (row1 < H) hi = clampi(row1 - r.mid_rows, lo, a1);
        }
        // which row ranges of this iteration the requested part runs
        int ranges[2][2], nr = 0;
        const bool before = r.mid < 0 || i < r.mid, at = r.mid >= 0 && i == r.mid;
        if (part == RMD_ATROUS_ALL || (part == RMD_ATROUS_HEAD && before) || (part == RMD_ATROUS_TAIL && !before && !at)) {
            ranges[nr][0] = a0; ranges[nr][1] = a1; ++nr;
        } else if (part == RMD_ATROUS_HEAD && at) {
            if (lo > a0) { ranges[nr][0] = a0; ranges[nr][1] = lo; ++nr; }
            if (a1 > hi) { ranges[nr][0] = hi; ranges[nr][1] = a1; ++nr; }
        } else if (part == RMD_ATROUS_INTERIOR && at) {
            if (hi > lo) { ranges[nr][0] = lo; ranges[nr][1] = hi; ++nr; }
        }
        if (nr == 2) {          // the two boundary bands of the exchanged iteration: one launch
            if (int e = rmd_svgf_atrous2(f, p, i, in, out, ranges[0][0], ranges[0][1], ranges[1][0], ranges[1][1], stream)) return e;
        } else if (nr == 1) {
            if (int e = rmd_svgf_atrous(f, p, i, in, out, ranges[0][0], ranges[0][1], stream)) return e;
        }
        if (nr == 0) continue;
        if (at && part == RMD_ATROUS_HEAD && hi > lo) continue;        // (iteration mid is complete only after its interior part)
        if (i == n - 1 && i == p->hist_iteration && f->hist_color_out != out) {
            const size_t off = (size_t)(a0 - f->buf_row0) * f->width * 4;
            RMD_HIP(hipMemcpyAsync(f->hist_color_out + off, out + off, (size_t)(a1 - a0) * f->width * 16,
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
        }
        // next frame's history (hist_color_out, and t_moments since T ran before) is complete here -- up to the rows a
        // mid-frame exchange still has to deliver when hist_iteration IS the exchanged iteration
        if (i == p->hist_iteration && history_ready_event)
            RMD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(history_ready_event), as_stream(stream)));
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
    c->nd[0] = c->nd[1] = nullptr; c->nd_cur = 0; c->nd_valid = false; c->t_debug = nullptr;
    c->tile_flags = nullptr;
    struct { void** p; size_t bytes; } planes[] = {
        { (void**)&c->hist_color[0], c->plane_bytes }, { (void**)&c->hist_color[1], c->plane_bytes },
        { (void**)&c->hist_moments[0], c->plane_bytes / 2 }, { (void**)&c->hist_moments[1], c->plane_bytes / 2 },
        { (void**)&c->hist_len[0], c->plane_bytes / 16 }, { (void**)&c->hist_len[1], c->plane_bytes / 16 },
        { (void**)&c->t_color, c->plane_bytes }, { (void**)&c->v_color, c->plane_bytes },
        { (void**)&c->ping[0], c->plane_bytes }, { (void**)&c->ping[1], c->plane_bytes },
        { (void**)&c->tile_flags, RMD_TILE_FLAGS_BYTES(width, height) } };
    for (auto& q : planes) *q.p = nullptr;
    for (auto& q : planes) {
        hipError_t e = hipMalloc(q.p, q.bytes);
        if (e == hipSuccess) e = hipMemset(*q.p, 0, q.bytes);
        if (e != hipSuccess) {
            for (auto& z : planes) if (*z.p) (void)hipFree(*z.p);
            delete c;
            return hip_fail(e, "rmd_svgf_context_create: hipMalloc");
        }
    }
    *out = c;
    return RMD_OK;
}

void rmd_svgf_context_destroy(rmd_svgf_context* c)
{
    if (!c) return;
    if (c->tile_flags) (void)hipFree(c->tile_flags);
    void* planes[] = { c->hist_color[0], c->hist_color[1], c->hist_moments[0], c->hist_moments[1], c->hist_len[0], c->hist_len[1],
                       c->t_color, c->v_color, c->ping[0], c->ping[1], c->nd[0], c->nd[1] };
    for (void* q : planes) if (q) (void)hipFree(q);
    delete c;
}

int rmd_svgf_context_reset_history(rmd_svgf_context* c, void* stream)
{
    if (!c) return fail(RMD_E_NULL, "rmd_svgf_context_reset_history: ctx is NULL");
    c->has_history = false;
    c->nd_valid = false;
    for (int i = 0; i < 2; ++i) {
        RMD_HIP(hipMemsetAsync(c->hist_color[i], 0, c->plane_bytes, as_stream(stream)));
        RMD_HIP(hipMemsetAsync(c->hist_moments[i], 0, c->plane_bytes / 2, as_stream(stream)));
        RMD_HIP(hipMemsetAsync(c->hist_len[i], 0, c->plane_bytes / 16, as_stream(stream)));
    }
    return RMD_OK;
}

int rmd_svgf_context_describe(rmd_svgf_context* c, rmd_svgf_frame_desc* f)
{
    if (!c || !f) return fail(RMD_E_NULL, "rmd_svgf_context_describe: NULL argument");
    f->width = c->width; f->height = c->height; f->buf_row0 = c->buf_row0; f->buf_rows = c->buf_rows;
    f->hist_color = c->has_history ? c->hist_color[c->cur] : nullptr;
    f->hist_moments = c->has_history ? c->hist_moments[c->cur] : nullptr;
    f->hist_len = c->has_history ? c->hist_len[c->cur] : nullptr;
    f->t_color = c->t_color;
    f->t_moments = c->hist_moments[c->cur ^ 1];
    f->t_len = c->hist_len[c->cur ^ 1];
    f->v_color = c->v_color;
    f->hist_color_out = c->hist_color[c->cur ^ 1];
    f->ping[0] = c->ping[0]; f->ping[1] = c->ping[1];
    f->v_tile_flags = c->tile_flags;
    return RMD_OK;
}

static int context_frame_desc(rmd_svgf_context* c, const float* color, const float* nd, const float* motion, const float* prev_nd,
                              float* out, rmd_svgf_frame_desc* f)
{
    if (!c) return fail(RMD_E_NULL, "rmd_svgf_context_denoise: ctx is NULL");
    if (!color || !nd || !motion || !out) return fail(RMD_E_NULL, "rmd_svgf_context_denoise: a required plane is NULL");
    *f = rmd_svgf_frame_desc{};
    if (int e = rmd_svgf_context_describe(c, f)) return e;
    f->color = color; f->nd = nd; f->motion = motion;
    // (after a frame of rmd_svgf_gbuffer_frame the previous nd is the context's own plane, which the caller cannot name:
    // the first float-plane frame behind it starts a new history)
    const bool use_hist = c->has_history && prev_nd != nullptr && !c->nd_valid;
    f->prev_nd = use_hist ? prev_nd : nullptr;
    if (!use_hist) { f->hist_color = nullptr; f->hist_moments = nullptr; f->hist_len = nullptr; }
    f->out_color = out;
    f->t_debug = c->t_debug; f->stats = nullptr;
    return RMD_OK;
}


int rmd_svgf_context_denoise(rmd_svgf_context* c, const rmd_svgf_params* p, const float* color, const float* nd,
                             const float* motion, const float* prev_nd, float* out, int row0, int row1, void* stream)
{
    rmd_svgf_frame_desc f;
    if (int e = context_frame_desc(c, color, nd, motion, prev_nd, out, &f)) return e;
    if (int e = rmd_svgf_frame(&f, p, row0, row1, stream)) return e;
    c->cur ^= 1;            // this frame's t_moments / hist_color_out become the history
    c->has_history = true;
    c->nd_valid = false;    // (the previous frame's nd is the caller's plane, not one of the context's)
    return RMD_OK;
}

int rmd_svgf_context_denoise_part(rmd_svgf_context* c, const rmd_svgf_params* p, const float* color, const float* nd,
                                  const float* motion, const float* prev_nd, float* out, int row0, int row1, void* stream, int part)
{
    rmd_svgf_frame_desc f;
    if (int e = context_frame_desc(c, color, nd, motion, prev_nd, out, &f)) return e;
    if (part < RMD_ATROUS_ALL || part > RMD_ATROUS_TAIL) return fail(RMD_E_PARAM, "rmd_svgf_context_denoise_part: part %d", part);
    // RMD_ATROUS_ALL is "everything in order" = rmd_svgf_context_denoise: T + V come first (they used to be skipped, so the
    // iterations ran on a stale v_color and the history was rotated over it)
    if (part == RMD_ATROUS_HEAD || part == RMD_ATROUS_ALL)
        if (int e = rmd_svgf_frame_tv(&f, p, row0, row1, stream)) return e;
    if (int e = rmd_svgf_frame_atrous_part(&f, p, row0, row1, stream, nullptr, part)) return e;
    if (part == RMD_ATROUS_TAIL || part == RMD_ATROUS_ALL) {
        c->cur ^= 1;
        c->has_history = true;
        c->nd_valid = false;
    }
    return RMD_OK;
}

int rmd_svgf_context_mid_plane(rmd_svgf_context* c, const rmd_svgf_params* p, float** plane)
{
    if (!c || !plane) return fail(RMD_E_NULL, "rmd_svgf_context_mid_plane: NULL argument");
    int mid[2];
    if (int e = rmd_svgf_frame_mid_exchange(p, mid)) return e;
    if (mid[0] < 0) return fail(RMD_E_PARAM, "rmd_svgf_context_mid_plane: exchange_iteration is -1");
    rmd_svgf_frame_desc f = {};
    if (int e = rmd_svgf_context_describe(c, &f)) return e;
    return rmd_svgf_frame_iteration_plane(&f, p, mid[0], plane);
}

int rmd_svgf_context_set_debug_plane(rmd_svgf_context* c, int* t_debug)
{
    if (!c) return fail(RMD_E_NULL, "rmd_svgf_context_set_debug_plane: ctx is NULL");
    if (!aligned_to(t_debug, 16)) return fail(RMD_E_ALIGN, "rmd_svgf_context_set_debug_plane: the int4 plane must be 16-byte aligned");
    c->t_debug = t_debug;
    return RMD_OK;
}

// One frame of SVGF on the reference's own frame descriptor (include/gbuffer.h:6-14): uchar4 render / albedo / normal in,
// uchar4 denoised out, six launches.  The 8-bit ends are fused into the first and the last of them (pixel_convert.h).
int rmd_svgf_gbuffer_frame(rmd_gbuffer frame, rmd_svgf_context* c, const rmd_svgf_params* p, const float* motion,
                           float albedo_eps, void* stream)
{
    if (!c || !p) return fail(RMD_E_NULL, "rmd_svgf_gbuffer_frame: ctx / params is NULL");
    if (frame.shape.x != c->width || frame.shape.y != c->height)
        return fail(RMD_E_SHAPE, "rmd_svgf_gbuffer_frame: GBuffer is %dx%d, the context %dx%d", frame.shape.x, frame.shape.y, c->width, c->height);
    if (c->buf_row0 != 0 || c->buf_rows != c->height)
        return fail(RMD_E_ROWS, "rmd_svgf_gbuffer_frame: the GBuffer holds whole frames; the context holds rows [%d,%d) of %d",
                    c->buf_row0, c->buf_row0 + c->buf_rows, c->height);
    if (!frame.render || !frame.albedo || !frame.normal || !frame.denoised)
        return fail(RMD_E_NULL, "rmd_svgf_gbuffer_frame: render / albedo / normal / denoised must all be set");
    if (frame.denoised == frame.render || frame.denoised == frame.albedo || frame.denoised == frame.normal)
        return fail(RMD_E_BUFFER, "rmd_svgf_gbuffer_frame: denoised aliases an input plane");
    Reach r;
    if (int e = compute_reach(p, r)) return e;
    if (r.mid >= 0) return fail(RMD_E_PARAM, "rmd_svgf_gbuffer_frame: whole frames only (exchange_iteration must be -1)");
    if (p->var_radius != 3 || p->tv_workgroups != 0 || p->var_h_threshold > 256)
        return fail(RMD_E_UNSUPPORTED, "rmd_svgf_gbuffer_frame: the 8-bit front end lives in the fused T+V launch (var_radius 3, "
                    "tv_workgroups 0, var_h_threshold <= 256); other settings take the float planes (rmd_convert_u8_to_f32, rmd_svgf_context_denoise)");
    for (int k = 0; k < 2; ++k)
        if (!c->nd[k]) {
            RMD_HIP(hipMalloc((void**)&c->nd[k], c->plane_bytes));
            RMD_HIP(hipMemsetAsync(c->nd[k], 0, c->plane_bytes, as_stream(stream)));
        }
    const int H = c->height, n = p->iterations;
    rmd_svgf_frame_desc f = {};
    if (int e = rmd_svgf_context_describe(c, &f)) return e;
    const bool use_hist = c->has_history && c->nd_valid;
    if (!use_hist) { f.hist_color = nullptr; f.hist_moments = nullptr; f.hist_len = nullptr; }
    f.color = nullptr; f.motion = motion;
    f.nd = c->nd[c->nd_cur];
    f.prev_nd = use_hist ? c->nd[c->nd_cur ^ 1] : nullptr;
    f.t_debug = c->t_debug; f.stats = nullptr;
    // the last iteration stores bytes; if it is also the history iteration its floats are needed too: it then writes
    // hist_color_out and a conversion launch follows (not the default: hist_iteration = 0 of 5)
    const bool last_is_hist = p->hist_iteration == n - 1;
    f.out_color = last_is_hist ? f.hist_color_out : nullptr;
    if ((n >= 2 + (last_is_hist ? 0 : 1) && !f.ping[0]) || !f.v_color || !f.hist_color_out)
        return fail(RMD_E_NULL, "rmd_svgf_gbuffer_frame: context planes missing");
    const GBuffer8 g8 = { reinterpret_cast<const uchar4*>(frame.render), reinterpret_cast<const uchar4*>(frame.albedo),
                          reinterpret_cast<const uchar4*>(frame.normal), reinterpret_cast<uchar4*>(frame.denoised), albedo_eps };
    if (int e = launch_temporal_variance(&f, p, 0, H, 0, H, stream, &g8)) return e;
    for (int i = 0; i < n; ++i) {
        const float* in;
        float* out;
        atrous_route(&f, p, i, &in, &out);
        const bool bytes = i == n - 1 && !last_is_hist;
        if (int e = launch_atrous(&f, p, i, in, out, 0, H, 0, 0, stream, nullptr, bytes ? &g8 : nullptr)) return e;
    }
    if (last_is_hist)
        if (int e = launch_modulate_to_u8(f.hist_color_out, frame.albedo, frame.denoised, (size_t)c->width * H, stream)) return e;
    c->cur ^= 1;
    c->has_history = true;
    c->nd_cur ^= 1;
    c->nd_valid = true;
    return RMD_OK;
}

int rmd_svgf_context_history(rmd_svgf_context* c, float** hist_color, float** hist_moments, unsigned char** hist_len)
{
    if (!c || !hist_color || !hist_moments || !hist_len) return fail(RMD_E_NULL, "rmd_svgf_context_history: NULL argument");
    *hist_color = c->hist_color[c->cur];
    *hist_moments = c->hist_moments[c->cur];
    *hist_len = c->hist_len[c->cur];
    return RMD_OK;
}

}  // extern "C"
