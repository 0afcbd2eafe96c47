// svgf_temporal.hip — T pass: temporal reprojection + history accumulation (SURVEY Appendix A.T).
//
// Not in the reference (README.md:3-10 names "temporal accumulation"; no code exists).  One
// thread per pixel; a wave owns 64 consecutive x so color / nd / motion are coalesced 16-byte
// (8-byte for motion) loads.  The four bilinear history taps are gathers: neighbouring pixels
// reproject to neighbouring history pixels, so each history line is fetched once from HBM and
// re-served by L1/L2.  Algorithmic traffic 120 B/px: HBM-bound, no LDS needed.
//
// Bit-exact contract: q0, the 4-bit tap mask and the history length use only +,-,*,compare,
// floor on fp32 in the oracle's operation order (this TU is built with -ffp-contract=off), so
// they match oracle/svgf_oracle.c:orc_svgf_temporal bit for bit; so do the float outputs
// (IEEE division, no transcendental functions in this pass).
#include "common.h"

namespace rmd {

struct TemporalArgs {
    Geom g;
    const float4* color; const float4* nd; const float2* motion;
    const float4* hist_color; const float4* hist_moments; const float4* prev_nd;
    float4* t_color; float4* t_moments; int4* t_debug;
    float4* v_color;          // optional second copy of t_color (fused frame: V then only rewrites short-history pixels)
    unsigned char* tile_flags; // optional: 1 per 64x4 tile (global tiling) holding a pixel with h < var_h_threshold
    int tiles_x, var_h_threshold;
    int sparse_t_color;        // fused frame with the tile form of V: t_color is written only inside flagged tiles (V reads
                               // every other pixel of its windows from v_color, which holds the same values there)
    int row0, row1;
    float alpha_color, alpha_moments, k_z, k_n;
    int h_max, max_motion_rows;
};

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + (b - a) * t; }

// One 64x4 tile of the GLOBAL tiling (rows 4k..4k+3, so T and V agree on tiles) by one workgroup.
__device__ __forceinline__ void temporal_tile(const TemporalArgs& a, const int tile_x, const int tile_y)
{
    const Geom g = a.g;
    const int x = tile_x * 64 + (threadIdx.x & 63);
    const int y = tile_y * 4 + (threadIdx.x >> 6);
    bool short_history = false;
    const bool active = x < g.W && y >= a.row0 && y < a.row1;
    float4 tc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    size_t i = 0;
    if (active) {
    i = pix_index(g, x, y);

    const float4 c = a.color[i];
    const float4 nd = a.nd[i];
    const float2 m = a.motion[i];

    // A.T.1
    const float qx = (float)x + m.x, qy = (float)y + m.y;
    const float fqx = floorf(qx), fqy = floorf(qy);
    const int q0x = (int)fqx, q0y = (int)fqy;
    const float fx = qx - fqx, fy = qy - fqy;
    const float wk[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };

    // A.T.2: depth gradient by forward differences clamped at the border
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    const float zr = a.nd[pix_index(g, x1, y)].w;
    const float zd = a.nd[pix_index(g, x, y1)].w;
    const float gz = fabsf(zr - nd.w) + fabsf(zd - nd.w);
    const float zthr = a.k_z * (gz + 1e-2f);
    const bool p_zero = is_zero3(nd);

    int mask = 0;
    float wsum = 0.0f, pcx = 0.0f, pcy = 0.0f, pcz = 0.0f, pm1 = 0.0f, pm2 = 0.0f;
    float best_w = -1.0f;
    int best_h = 0;
    if (a.prev_nd) {                               // NULL = no history yet (first frame / after a reset)
        // All twelve history gathers are issued up front, at tap coordinates clamped into the rows
        // the planes hold, and validity is decided afterwards: a dependent load-test-load chain per
        // tap keeps too few bytes in flight for an HBM-bound pass.  Same arithmetic, same order.
        const int ylo = max(max(g.buf_row0, 0), y - a.max_motion_rows);
        const int yhi = min(min(g.buf_row0 + g.buf_rows, g.H) - 1, y + a.max_motion_rows);
        size_t ti[4];
        bool inb[4];
        float4 pn[4], hc[4], hm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = q0x + (k & 1), ty = q0y + (k >> 1);
            inb[k] = tx >= 0 && tx < g.W && ty >= 0 && ty < g.H && abs(ty - y) <= a.max_motion_rows;
            ti[k] = pix_index(g, min(max(tx, 0), g.W - 1), min(max(ty, ylo), yhi));
        }
        // Two batches: the four prev_nd gathers decide tap validity; only then the eight
        // hist_color / hist_moments gathers are issued.  The fence keeps the register footprint at
        // <= 56 VGPRs, so that one wave of this kernel fits on a SIMD beside three a-trous waves
        // (3 x 152 of 512 registers) when frames are pipelined over two streams.
#pragma unroll
        for (int k = 0; k < 4; ++k) pn[k] = a.prev_nd[ti[k]];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok_n = p_zero ? is_zero3(pn[k]) : ((pn[k].x * nd.x + pn[k].y * nd.y + pn[k].z * nd.z) >= a.k_n);
            ok[k] = inb[k] && (fabsf(pn[k].w - nd.w) <= zthr) && ok_n;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { hc[k] = a.hist_color[ti[k]]; hm[k] = a.hist_moments[ti[k]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            mask |= 1 << k;
            const float w = wk[k];
            wsum += w;
            pcx += w * hc[k].x; pcy += w * hc[k].y; pcz += w * hc[k].z;
            pm1 += w * hm[k].x; pm2 += w * hm[k].y;
            if (w > best_w) { best_w = w; best_h = (int)hm[k].z; }
        }
    }

    // A.T.3
    int h;
    if (mask != 0 && wsum >= 0.01f) {
        pcx /= wsum; pcy /= wsum; pcz /= wsum; pm1 /= wsum; pm2 /= wsum;
        h = min(best_h + 1, a.h_max);
        h = max(h, 1);
    } else {
        h = 1;
        pcx = pcy = pcz = 0.0f; pm1 = pm2 = 0.0f;
    }

    // A.T.4
    const float inv_h = 1.0f / (float)h;
    const float a_c = a.alpha_color > inv_h ? a.alpha_color : inv_h;
    const float a_m = a.alpha_moments > inv_h ? a.alpha_moments : inv_h;
    const float l = lum3(c.x, c.y, c.z);
    const float m1 = lerpf(pm1, l, a_m), m2 = lerpf(pm2, l * l, a_m);
    float var = m2 - m1 * m1;
    if (!(var > 0.0f)) var = 0.0f;

    tc = make_float4(lerpf(pcx, c.x, a_c), lerpf(pcy, c.y, a_c), lerpf(pcz, c.z, a_c), var);
    if (!a.sparse_t_color) a.t_color[i] = tc;
    if (a.v_color) a.v_color[i] = tc;
    a.t_moments[i] = make_float4(m1, m2, (float)h, 0.0f);
    if (a.t_debug) a.t_debug[i] = make_int4(q0x, q0y, mask, h);
    short_history = h < a.var_h_threshold;
    }
    if (a.tile_flags) {
        const int any = __syncthreads_or(short_history ? 1 : 0);
        if (threadIdx.x == 0) a.tile_flags[(size_t)tile_y * a.tiles_x + tile_x] = (unsigned char)(any != 0);
        // 16 of T's 136 B per pixel: in the steady state ~2 % of the tiles are flagged
        if (a.sparse_t_color && any && active) a.t_color[i] = tc;
    }
}

// one workgroup per tile: the pass on its own runs at HBM speed this way
__global__ __launch_bounds__(256) void svgf_temporal_kernel(TemporalArgs a)
{
    temporal_tile(a, blockIdx.x, a.row0 / 4 + blockIdx.y);
}

#ifdef RMD_EXPERIMENTS
// A fixed, small number of workgroups (one per CU) that walk over the tiles: the form that runs
// UNDERNEATH the a-trous launches of the previous frame (rmd_svgf_params.tv_workgroups).  Three
// a-trous workgroups leave 56 VGPRs per SIMD; one wave of this kernel fits there.  A grid of one
// short workgroup per tile fits there too, but whenever an a-trous launch retires, its pending
// workgroups refill the freed registers faster than the next a-trous launch can claim them, and
// that launch then waits for the whole T pass (measured: 385 us instead of 135).  A persistent
// grid has a constant footprint, so the next a-trous launch finds its three slots per CU.
// EXPERIMENTAL and off by default: with 256 workgroups the pass takes 410 us on an idle GPU (one
// wave per SIMD, ~3.3 us per tile of dependent gathers) and 565 us underneath a-trous launches,
// which it slows from 130 to ~215 us each: together the two saturate HBM (3.5 + 2 TB/s) and the
// a-trous ring refill stops being hidden.  Frames come out at 1.10 ms instead of 0.94 ms serial.
__global__ __launch_bounds__(256) void svgf_temporal_persistent_kernel(TemporalArgs a, int tiles_y)
{
    // few instructions, long latencies: issue them ahead of the a-trous waves (which run at 3..0)
    __builtin_amdgcn_s_setprio(3);
    const int ntiles = a.tiles_x * tiles_y;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) temporal_tile(a, t % a.tiles_x, a.row0 / 4 + t / a.tiles_x);
}
#endif

}  // namespace rmd

using namespace rmd;

int rmd::launch_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream, bool fused,
                         bool sparse_t_color)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_temporal: params is NULL");
    if (!f->color || !f->nd || !f->motion || !f->t_color || !f->t_moments)
        return fail(RMD_E_NULL, "rmd_svgf_temporal: a required plane is NULL");
    // hist_color / hist_moments / prev_nd all NULL = "no history": every pixel is a disocclusion
    const bool has_hist = f->hist_color && f->hist_moments && f->prev_nd;
    if (!has_hist && (f->hist_color || f->hist_moments || f->prev_nd))
        return fail(RMD_E_NULL, "rmd_svgf_temporal: history planes must be all set or all NULL");
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_temporal: rows [%d,%d) invalid", row0, row1);
    if (p->max_motion_rows < 0 || p->h_max < 1) return fail(RMD_E_PARAM, "rmd_svgf_temporal: max_motion_rows/h_max invalid");
    if (p->tv_workgroups < 0 || p->tv_workgroups > 65536) return fail(RMD_E_PARAM, "rmd_svgf_temporal: tv_workgroups %d outside [0,65536]", p->tv_workgroups);
    // current-frame planes: +1 row (depth gradient); history planes: +-(max_motion_rows) rows
    if (int e = check_rows_in_buffer(f, row0, row1 + 1, "rmd_svgf_temporal (current frame)")) return e;
    if (has_hist)
        if (int e = check_rows_in_buffer(f, row0 - p->max_motion_rows, row1 + p->max_motion_rows, "rmd_svgf_temporal (history)")) return e;
    const void* planes16[] = { f->color, f->nd, f->hist_color, f->hist_moments, f->prev_nd, f->t_color, f->t_moments, f->t_debug };
    for (const void* q : planes16)
        if (!aligned_to(q, 16)) return fail(RMD_E_ALIGN, "rmd_svgf_temporal: float4 planes must be 16-byte aligned");
    if (!aligned_to(f->motion, 8)) return fail(RMD_E_ALIGN, "rmd_svgf_temporal: motion must be 8-byte aligned");

    TemporalArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.color = (const float4*)f->color; a.nd = (const float4*)f->nd; a.motion = (const float2*)f->motion;
    a.hist_color = (const float4*)f->hist_color; a.hist_moments = (const float4*)f->hist_moments;
    a.prev_nd = (const float4*)f->prev_nd;
    a.t_color = (float4*)f->t_color; a.t_moments = (float4*)f->t_moments; a.t_debug = (int4*)f->t_debug;
    a.v_color = nullptr;
    a.sparse_t_color = (fused && sparse_t_color) ? 1 : 0;
    a.tile_flags = nullptr; a.tiles_x = (f->width + 63) / 64; a.var_h_threshold = p->var_h_threshold;
    if (fused) {
        a.tile_flags = f->v_tile_flags;
        if (!f->v_color || !aligned_to(f->v_color, 16) || f->v_color == f->t_color)
            return fail(RMD_E_NULL, "rmd_svgf_temporal: fused v_color plane is NULL, misaligned or aliases t_color");
        a.v_color = (float4*)f->v_color;
    }
    a.row0 = row0; a.row1 = row1;
    a.alpha_color = p->alpha_color; a.alpha_moments = p->alpha_moments; a.k_z = p->k_z; a.k_n = p->k_n;
    a.h_max = p->h_max; a.max_motion_rows = p->max_motion_rows;
    dim3 grid((f->width + 63) / 64, (row1 - 1) / 4 - row0 / 4 + 1);
    if (p->tv_workgroups > 0) {
#ifdef RMD_EXPERIMENTS
        hipLaunchKernelGGL(svgf_temporal_persistent_kernel, dim3(p->tv_workgroups), dim3(256), 0, as_stream(stream), a, (int)grid.y);
#else
        return fail(RMD_E_UNSUPPORTED, "rmd_svgf_temporal: tv_workgroups > 0 (persistent T / V grids) is an experiment (make experiments)");
#endif
    } else {
        hipLaunchKernelGGL(svgf_temporal_kernel, grid, dim3(256), 0, as_stream(stream), a);
    }
    RMD_LAUNCH_CHECK("svgf_temporal_kernel");
    return RMD_OK;
}

extern "C" int rmd_svgf_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream)
{
    return rmd::launch_temporal(f, p, row0, row1, stream, /*fused=*/false, /*sparse_t_color=*/false);
}
