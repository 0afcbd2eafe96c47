// svgf_tv.h -- device code shared by the T pass (svgf_temporal.hip), the V pass (svgf_variance.hip) and the fused T+V
// kernel of whole frames: Appendix A.T for one pixel, and the 7x7 spatial estimate of Appendix A.V on a tile + halo
// staged in LDS.  One definition each, so every launch form produces the same bits.
#pragma once
#include "common.h"
#include "pixel_convert.h"

namespace rmd {

struct TemporalArgs {
    Geom g;
    const float4* color; const float4* nd; const float2* motion;
    const float4* hist_color; const float2* hist_moments; const unsigned char* hist_len; const float4* prev_nd;
    float4* t_color; float2* t_moments; unsigned char* t_len; int4* t_debug;
    float4* v_color;          // optional second copy of t_color (fused frame: V then only rewrites short-history pixels)
    unsigned char* tile_flags; // optional: 1 per 64x4 tile (global tiling) holding a pixel with h < var_h_threshold
    int tiles_x, var_h_threshold;
    int sparse_t_color;        // fused frame with the tile form of V: t_color is written only inside flagged tiles (V reads
                               // every other pixel of its windows from v_color, which holds the same values there)
    int row0, row1;
    float alpha_color, alpha_moments, k_z, k_n;
    int h_max, max_motion_rows;
    // 8-bit front end (rmd_svgf_gbuffer_frame, the IN8 instantiations): color / nd are not read -- the inputs are the uchar4
    // planes of the reference's GBuffer (include/gbuffer.h:9-12), converted and demodulated in registers with the functions of
    // the standalone conversion kernels (pixel_convert.h) -- and the float nd plane is WRITTEN, once, for the a-trous passes
    // and as the next frame's prev_nd.  motion may be NULL in either form: a static camera.
    const uchar4* render8; const uchar4* albedo8; const uchar4* normal8;
    float4* nd_out;
    float albedo_eps;
};

// The temporal pass of the NEXT frame as a side job of the a-trous launches of this one (svgf_atrous.hip,
// rmd_svgf_frame_atrous_next): its tiles, row major over `t`'s rows, are claimed from `counter`.
struct AtrousSide {
    TemporalArgs t;
    unsigned* counter;     // device word, zero before the first launch that carries the job
    int units;             // tiles_x * tile rows of [t.row0, t.row1)
    int every;             // one tile per workgroup every `every` steps
};

// validates the T call and fills the kernel arguments (launch_temporal's; `fused`: v_color + tile flags as in rmd_svgf_frame_tv)
int make_temporal_args(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, bool fused, bool sparse_t_color,
                       TemporalArgs* out);
// the tiles of `side` nobody has claimed yet (a grid of workgroups that claim until the counter runs out)
int launch_temporal_claim(const AtrousSide& side, void* stream);
// g8 != NULL: the iteration stores modulated, quantised bytes to g8->denoised (pixel_convert.h) instead of floats to `out`
int launch_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration, const float* in, float* out,
                  int row0, int row1, int row0b, int row1b, void* stream, const AtrousSide* side, const GBuffer8* g8 = nullptr);
unsigned* side_counter_on_device();        // one zero-initialised word per device (runtime.hip)

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + (b - a) * t; }

// Appendix A.T for ONE pixel (x, y) inside the frame and the buffer: returns c' + variance in `tc`, the moments (m1', m2') in
// `mom` (the history length h is dbg.w), the bit-exact integer outputs in `dbg`, the pixel's (normal, depth) in `nd`.  Stores nothing.
// The history planes are read as they are stored: hist_color float4 (its variance channel is not used: T recomputes the variance
// from the moments), hist_moments float2, hist_len one byte -- 16 + 8 + 1 bytes per tap instead of the 16 + 16 of a float4
// (m1, m2, h, 0) plane.
// IN8: illumination and (normal, depth) come from the GBuffer's uchar4 planes -- rmd_convert_u8_to_f32 (render, albedo: c/255;
// normal: c/255 renormalised, w/255 = depth) and rmd_demodulate in registers, the operations of those kernels in their order.
template <bool IN8 = false>
__device__ __forceinline__ void temporal_pixel(const TemporalArgs& a, const int x, const int y, float4& tc, float2& mom, int4& dbg, float4& nd)
{
    const Geom g = a.g;
    const size_t i = pix_index(g, x, y);

    float4 c;
    if constexpr (IN8) {
        c = demodulated(float4_from_u8(a.render8[i], false, 0.0f), float4_from_u8(a.albedo8[i], false, 0.0f), a.albedo_eps);
        nd = float4_from_u8(a.normal8[i], true, -1.0f);
    } else {
        c = a.color[i];
        nd = a.nd[i];
    }
    const float2 m = a.motion ? a.motion[i] : make_float2(0.0f, 0.0f);

    // A.T.1
    const float qx = (float)x + m.x, qy = (float)y + m.y;
    const float fqx = floorf(qx), fqy = floorf(qy);
    const int q0x = (int)fqx, q0y = (int)fqy;
    const float fx = qx - fqx, fy = qy - fqy;
    const float wk[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };

    // A.T.2: depth gradient by forward differences clamped at the border
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    float zr, zd;
    if constexpr (IN8) {
        zr = unit_from_u8(reinterpret_cast<const unsigned char*>(a.normal8 + pix_index(g, x1, y))[3]);
        zd = unit_from_u8(reinterpret_cast<const unsigned char*>(a.normal8 + pix_index(g, x, y1))[3]);
    } else {
        zr = a.nd[pix_index(g, x1, y)].w;
        zd = a.nd[pix_index(g, x, y1)].w;
    }
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
        float4 pn[4], hc[4];
        float2 hm[4];
        unsigned char hl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = q0x + (k & 1), ty = q0y + (k >> 1);
            inb[k] = tx >= 0 && tx < g.W && ty >= 0 && ty < g.H && abs(ty - y) <= a.max_motion_rows;
            ti[k] = pix_index(g, min(max(tx, 0), g.W - 1), min(max(ty, ylo), yhi));
        }
        // Two batches: the four prev_nd gathers decide tap validity; only then the eight
        // hist_color / hist_moments gathers are issued (the fence keeps the register footprint at <= 56 VGPRs).
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
        for (int k = 0; k < 4; ++k) { hc[k] = a.hist_color[ti[k]]; hm[k] = a.hist_moments[ti[k]]; hl[k] = a.hist_len[ti[k]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            mask |= 1 << k;
            const float w = wk[k];
            wsum += w;
            pcx += w * hc[k].x; pcy += w * hc[k].y; pcz += w * hc[k].z;
            pm1 += w * hm[k].x; pm2 += w * hm[k].y;
            if (w > best_w) { best_w = w; best_h = (int)hl[k]; }
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
    mom = make_float2(m1, m2);
    dbg = make_int4(q0x, q0y, mask, h);
}

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
        float4 nd;
        float2 mom;
        int4 dbg;
        temporal_pixel(a, x, y, tc, mom, dbg, nd);
        if (!a.sparse_t_color) a.t_color[i] = tc;
        if (a.v_color) a.v_color[i] = tc;
        a.t_moments[i] = mom;
        a.t_len[i] = (unsigned char)dbg.w;
        if (a.t_debug) a.t_debug[i] = dbg;
        short_history = dbg.w < a.var_h_threshold;
    }
    if (a.tile_flags) {
        const int any = __syncthreads_or(short_history ? 1 : 0);
        if (threadIdx.x == 0) a.tile_flags[(size_t)tile_y * a.tiles_x + tile_x] = (unsigned char)(any != 0);
        // 16 of T's 136 B per pixel: in the steady state ~2 % of the tiles are flagged
        if (a.sparse_t_color && any && active) a.t_color[i] = tc;
    }
}

constexpr int kVR = 3, kVW = 64 + 2 * kVR, kVH = 4 + 2 * kVR;      // a 64x4 tile + the 3-pixel reach of the 7x7 window: 70 x 10

// Appendix A.V for the pixel (x, y) = tile pixel (lx, ly): the 49 taps read T's output (rgb) and nd of the staged 70 x 10 region
// through `rgb(ry, rx)` / `ndz(ry, rx)` (float4 each; rgb's .w is not used).  Tap order dx outer / dy inner, the arithmetic of
// variance_pixel (svgf_variance.hip): identical bits.  `passthrough` is set when the weights vanish (sw < 1e-10): the pixel then
// keeps T's own value, variance included, which the caller supplies (the result's xyz are T's colour already).
template <class Rgb, class Ndz>
__device__ __forceinline__ float4 variance_window_lds(Rgb rgb, Ndz ndz, const int lx, const int ly, const int x, const int y, const Geom& g,
                                                      const float sigma_n, const float sigma_z, const int h, bool& passthrough)
{
    const float4 c = rgb(ly + kVR, lx + kVR);
    const float4 nd = ndz(ly + kVR, lx + kVR);
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    const float gz = fabsf(ndz(ly + kVR, x1 - x + lx + kVR).w - nd.w) + fabsf(ndz(y1 - y + ly + kVR, lx + kVR).w - nd.w);
    const float za = sigma_z * fmaxf(gz, 1e-8f);
    const bool p_zero = is_zero3(nd);
    float sw = 0.0f, scx = 0.0f, scy = 0.0f, scz = 0.0f, sl = 0.0f, sl2 = 0.0f;
    // both loops unrolled: tap lengths become constants (nine distinct reciprocals instead of 48 square
    // roots and divisions) and the LDS reads of neighbouring taps overlap -- the few lanes that get
    // here are a latency chain, not a throughput problem
#pragma unroll
    for (int dx = -kVR; dx <= kVR; ++dx) {
#pragma unroll
        for (int dy = -kVR; dy <= kVR; ++dy) {
            const int tx = x + dx, ty = y + dy;
            if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) continue;
            const float4 tc = rgb(ly + kVR + dy, lx + kVR + dx);
            const float4 tn = ndz(ly + kVR + dy, lx + kVR + dx);
            float e;
            const bool t_zero = is_zero3(tn);
            if (p_zero || t_zero) {
                e = (p_zero && t_zero) ? 0.0f : kNegInf;
            } else {
                const float d = __builtin_fmaf(nd.z, tn.z, __builtin_fmaf(nd.y, tn.y, nd.x * tn.x));
                e = sigma_n * fast_log2(fmaxf(d, 0.0f));
            }
            if (dx != 0 || dy != 0) {
                const float len = sqrtf((float)(dx * dx + dy * dy));
                e = __builtin_fmaf(-fabsf(nd.w - tn.w), kLog2e / (za * len + 1e-8f), e);
            }
            const float w = fast_exp2(e);
            const float tl = lum3(tc.x, tc.y, tc.z);
            sw += w;
            scx = __builtin_fmaf(w, tc.x, scx); scy = __builtin_fmaf(w, tc.y, scy); scz = __builtin_fmaf(w, tc.z, scz);
            sl = __builtin_fmaf(w, tl, sl); sl2 = __builtin_fmaf(w, tl * tl, sl2);
        }
    }
    float4 o = c;
    passthrough = sw < 1e-10f;
    if (!passthrough) {
        const float el = sl / sw, el2 = sl2 / sw;
        float var = el2 - el * el;
        if (!(var > 0.0f)) var = 0.0f;
        var *= 4.0f / (float)max(h, 1);
        o = make_float4(scx / sw, scy / sw, scz / sw, var);
    }
    return o;
}

}  // namespace rmd
