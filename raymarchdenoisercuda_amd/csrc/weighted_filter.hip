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
#include <type_traits>
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

// ---- CROSS / WAVELET at spacing 1, 5x5 window: tile in LDS, two pixels per thread -------------------------------------------
// weighted_filter_kernel is VALU-bound (~36 instructions + v_exp per tap, one pixel per thread, every tap converted from
// bytes again).  Here a thread owns TWO pixels of a 64 x 8 tile, A in row y and B in row y + 4 of the same column, and every
// tap (dx, dy) is weighed for both at once with packed f32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma): lane 0 of the packed
// registers is A's window, lane 1 is B's.  The planes are converted to float once per staged pixel, and the LDS image is laid
// out FOR the packed operands: entry [r][c] of a plane holds the values of region rows r and r + 4 side by side
// ({rA, rB, gA, gB} and {bA, bB}), so that a ds_read_b128 + a ds_read_b64 land both pixels' tap in aligned register pairs
// (rows 4..7 of the 12-row region are stored twice: as the B half of entry r - 4 and the A half of entry r).
// Squared distances from packed differences: integers below 2^24, hence exactly the oracle's value.  (Staging |t|^2 beside
// the blue pair and forming |k|^2 + |t|^2 - 2 k.t saves one instruction per plane and tap and costs a third more LDS bytes:
// 153 us against 137 for CROSS at 4K -- the LDS pipe, not the VALU, is what the last step came from.)  Exponent and sums use
// the oracle's fused multiply-adds in its order (dx outer, dy inner; colour, albedo, normal), so the only difference to the
// oracle stays v_exp_f32 against expf, as in weighted_filter_kernel.  Out-of-frame taps get weight 0, which adds exactly
// nothing.  LDS 13 KB per plane (39 KB with all three; the VGPR budget keeps three workgroups per CU); the next tap's reads
// are issued before the current tap is weighed.
typedef float wf2 __attribute__((ext_vector_type(2)));

// Dilated WAVELET levels (spacing S = 2, 4, 8) are the same filter on the S row lattices y mod S: a workgroup takes 2 D
// consecutive rows of ONE lattice (pairs j and j + D, D = 4), the region is those rows + 2 lattice rows either side and the
// tile's 64 columns + 2 S either side, taps at column offsets dx S.  Regions of 41 and 46 KB (S = 2, 4: three workgroups per CU,
// 143-146 us per 4K level) and 55 KB (S = 8: two, 175 us) against 132 at S = 1 and 208 for the gather kernel.  At S = 16 the
// region is as wide again as the tile (74 KB, two workgroups per CU: 222 us; D = 8 with 512 threads 270 us with the older
// 32-byte entries): that level and the ones above stay on the gather kernel.  (Columns on the lattice as well -- the tile as 64 x 8 pixels of one of S^2
// sub-images, a 68-column region at every spacing -- was built and measured: 406 us at S = 2, 1115 us at S = 16; lanes that
// load and store 4 bytes every 4 S bytes cost more than all the arithmetic.)
template <bool HAS_A, bool HAS_N, bool WAVELET, int S, int D>
__global__ __launch_bounds__(64 * D, (S <= 4 ? 3 : 2)) void weighted_tile_kernel(WeightedArgs a)
{
    constexpr int TW = 64, R = 2, RW = TW + 2 * R * S, NR = D + 2 * R, NP = 1 + (HAS_A ? 1 : 0) + (HAS_N ? 1 : 0);
    constexpr int FULL = RW / 64, REM = RW - 64 * FULL;                 // whole 64-column chunks of the region, and the rest (a power of two)
    extern __shared__ __attribute__((aligned(16))) float4 weighted_lds[];
    float4* const prg = weighted_lds;                                    // [NP][NR][RW]  { rA, rB, gA, gB }
    float2* const pb2 = reinterpret_cast<float2*>(weighted_lds + NP * NR * RW);      // [NP][NR][RW]  { bA, bB }
    auto at = [](const int p, const int r, const int c) { return (p * NR + r) * RW + c; };
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int group = blockIdx.y / S, lattice = blockIdx.y - group * S;
    const int x0 = blockIdx.x * TW, y0 = group * (2 * D * S) + lattice;  // the tile's rows: y0 + j S, j = 0 .. 2 D - 1
    const uchar4* const planes[3] = { a.in, HAS_A ? a.albedo : a.normal, a.normal };          // in the oracle's order of terms
    const float inv2s[3] = { a.inv2s_color, HAS_A ? a.inv2s_albedo : a.inv2s_normal, a.inv2s_normal };
    auto put = [&](const int p, const int r, const int c, const uchar4 u, const uchar4 v) {      // entry r: region rows r (A half) and r + D (B half)
        const wf2 rp = { (float)u.x, (float)v.x }, gp = { (float)u.y, (float)v.y }, bp = { (float)u.z, (float)v.z };
        prg[at(p, r, c)] = make_float4(rp.x, rp.y, gp.x, gp.y);
        pb2[at(p, r, c)] = make_float2(bp.x, bp.y);
    };
    auto row_of = [&](const int r) { return (size_t)min(max(y0 + (r - R) * S, 0), a.H - 1) * a.W; };        // region row r
    auto col_of = [&](const int c) { return min(max(x0 - R * S + c, 0), a.W - 1); };                        // region column c
    {   // the whole chunks: a thread reads region rows wv, wv + D (and wv + 2 D) of its column and writes entries wv (and wv + D)
        const size_t r0 = row_of(wv), r1 = row_of(wv + D), r2 = row_of(wv + 2 * D);
#pragma unroll
        for (int i = 0; i < FULL; ++i) {
            const int c = lx + 64 * i, gx = col_of(c);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uchar4 u0 = planes[p][r0 + gx], u1 = planes[p][r1 + gx];
                put(p, wv, c, u0, u1);
                if (wv < 2 * R) put(p, wv + D, c, u1, planes[p][r2 + gx]);
            }
        }
    }
    if (REM > 0)
        for (int q = threadIdx.x; q < REM * NR; q += 64 * D) {           // the remaining columns, every entry: as few waves as hold them
            const int r = q / (REM > 0 ? REM : 1), c = 64 * FULL + (q - r * REM), gx = col_of(c);
            const size_t rA = row_of(r), rB = row_of(r + D);
#pragma unroll
            for (int p = 0; p < NP; ++p) put(p, r, c, planes[p][rA + gx], planes[p][rB + gx]);
        }
    __syncthreads();
    const int x = x0 + lx, yA = y0 + wv * S, yB = yA + D * S;
    if (x >= a.W || yA >= a.H) return;
    const bool inside = x0 - R * S >= 0 && x0 + TW + R * S <= a.W && y0 - R * S >= 0 && y0 + (2 * D - 1 + R) * S < a.H;     // no tap of the tile leaves the frame
    const int cx = lx + R * S;
    struct Tap { float4 rg[NP]; float2 b[NP]; };
    auto fetch = [&](const int dx, const int dyi) {              // window offset (dx S, (dyi - 2) S) of both pixels: entry row wv + dyi
        Tap t;
#pragma unroll
        for (int p = 0; p < NP; ++p) { t.rg[p] = prg[at(p, wv + dyi, cx + dx * S)]; t.b[p] = pb2[at(p, wv + dyi, cx + dx * S)]; }
        return t;
    };
    const Tap ctr = fetch(0, 2);
    wf2 kr[NP], kg[NP], kb[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { kr[p] = wf2{ ctr.rg[p].x, ctr.rg[p].y }; kg[p] = wf2{ ctr.rg[p].z, ctr.rg[p].w }; kb[p] = wf2{ ctr.b[p].x, ctr.b[p].y }; }
    wf2 sr = { 0, 0 }, sg = sr, sb = sr, sw = sr;
    auto window = [&](auto inside_c) {
        constexpr bool interior = decltype(inside_c)::value;
        constexpr float spline[3] = { 0.375f, 0.25f, 0.0625f };
        Tap nxt = fetch(-R, 0);
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int dx = t / 5 - R, dyi = t % 5, dy = dyi - R;
            const Tap cur = nxt;
            if (t + 1 < 25) nxt = fetch((t + 1) / 5 - R, (t + 1) % 5);
            __builtin_amdgcn_sched_barrier(0);               // the next tap's reads stay ahead of this tap's arithmetic (deeper: no gain)
            wf2 e = { 0, 0 };
            if (!WAVELET) { const float e0 = (float)(dx * dx + dy * dy) * a.inv2s_space; e = wf2{ e0, e0 }; }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const wf2 dr = kr[p] - wf2{ cur.rg[p].x, cur.rg[p].y }, dg = kg[p] - wf2{ cur.rg[p].z, cur.rg[p].w }, db = kb[p] - wf2{ cur.b[p].x, cur.b[p].y };
                const wf2 d2 = __builtin_elementwise_fma(db, db, __builtin_elementwise_fma(dg, dg, dr * dr));     // exact: integers < 2^24
                e = __builtin_elementwise_fma(d2, wf2{ inv2s[p], inv2s[p] }, e);
            }
            const wf2 m = e * wf2{ -1.442695041f, -1.442695041f };            // __expf(-e) = v_exp_f32(-e * log2(e)): the same product
            wf2 w = { __builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y) };
            if (WAVELET) { const float k = spline[dx < 0 ? -dx : dx] * spline[dy < 0 ? -dy : dy]; w = wf2{ k, k } * w; }
            if (!interior) {
                const bool colv = x + dx * S >= 0 && x + dx * S < a.W;
                if (!(colv && yA + dy * S >= 0 && yA + dy * S < a.H)) w.x = 0.0f;
                if (!(colv && yB + dy * S >= 0 && yB + dy * S < a.H)) w.y = 0.0f;
            }
            sr = __builtin_elementwise_fma(w, wf2{ cur.rg[0].x, cur.rg[0].y }, sr);
            sg = __builtin_elementwise_fma(w, wf2{ cur.rg[0].z, cur.rg[0].w }, sg);
            sb = __builtin_elementwise_fma(w, wf2{ cur.b[0].x, cur.b[0].y }, sb);
            sw += w;
        }
    };
    if (inside) window(std::true_type{}); else window(std::false_type{});
    // The six quotients s / sw as the compiler expands an IEEE division (reciprocal, one Newton step, two residual corrections
    // of the quotient), packed for the two pixels and with the reciprocal shared by the three channels; without the operand
    // scaling of v_div_scale / v_div_fixup, which is the identity here: the centre tap has weight k(0,0) > 0 and a weight is
    // at most 1, so sw lies in [2^-9, 25] and the sums in [0, 6375].  Same bits as `/` (the 4K test against the gather kernel).
    wf2 rcp = { __builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y) };
    rcp = __builtin_elementwise_fma(__builtin_elementwise_fma(-sw, rcp, wf2{ 1.0f, 1.0f }), rcp, rcp);
    auto quotient = [&](const wf2 n) {
        wf2 q = n * rcp;
        q = __builtin_elementwise_fma(__builtin_elementwise_fma(-sw, q, n), rcp, q);
        return __builtin_elementwise_fma(__builtin_elementwise_fma(-sw, q, n), rcp, q);
    };
    const wf2 qr = quotient(sr), qg = quotient(sg), qb = quotient(sb);
    a.out[(size_t)yA * a.W + x] = make_uchar4((unsigned char)qr.x, (unsigned char)qg.x, (unsigned char)qb.x, 0);
    if (yB < a.H)
        a.out[(size_t)yB * a.W + x] = make_uchar4((unsigned char)qr.y, (unsigned char)qg.y, (unsigned char)qb.y, 0);
}

constexpr int kTileMaxStep = 8;

template <bool HAS_A, bool HAS_N, bool WAVELET, int S, int D>
static int launch_weighted_tile_at(const WeightedArgs& a, hipStream_t stream)
{
    constexpr int NP = 1 + (HAS_A ? 1 : 0) + (HAS_N ? 1 : 0), RW = 64 + 4 * S, NR = D + 4;
    constexpr int lds_bytes = 24 * NP * NR * RW;
    static_assert(lds_bytes <= 160 * 1024, "region larger than a CU's LDS");
    const auto kernel = &weighted_tile_kernel<HAS_A, HAS_N, WAVELET, S, D>;
    if (first_use_on_device(reinterpret_cast<const void*>(kernel)))
        RMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const dim3 grid((a.W + 63) / 64, ((a.H + 2 * D * S - 1) / (2 * D * S)) * S);
    hipLaunchKernelGGL(kernel, grid, dim3(64 * D), lds_bytes, stream, a);
    RMD_LAUNCH_CHECK("weighted_tile_kernel");
    return RMD_OK;
}

template <bool HAS_A, bool HAS_N>
static int launch_weighted_tile(const WeightedArgs& a, hipStream_t stream)
{
    if (a.mode != RMD_FILTER_WAVELET) return launch_weighted_tile_at<HAS_A, HAS_N, false, 1, 4>(a, stream);
    switch (a.step) {
        case 1:  return launch_weighted_tile_at<HAS_A, HAS_N, true, 1, 4>(a, stream);
        case 2:  return launch_weighted_tile_at<HAS_A, HAS_N, true, 2, 4>(a, stream);
        case 4:  return launch_weighted_tile_at<HAS_A, HAS_N, true, 4, 4>(a, stream);
        default: return launch_weighted_tile_at<HAS_A, HAS_N, true, 8, 4>(a, stream);
    }
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

// Radii 13 .. 127: one thread per pixel, the SAME operations in the same order (per window row the horizontal sum over the in-frame dx,
// then the vertical accumulation over the in-frame dy, fused multiply-adds; hw, vw plain sums; one product, three divisions), so the
// same bits as the separable kernel would give -- without its LDS tile, at (2r+1)^2 gathers per pixel.  AVERAGE takes any radius (the
// reference's loop does, src/filter.cu:34); this keeps GAUSSIAN from refusing what AVERAGE accepts.
constexpr int kGaussDirectMaxRadius = 127;
struct GaussDirectArgs {
    const uchar4* in; uchar4* out;
    int W, H, radius;
    float g[kGaussDirectMaxRadius + 1];
};

__global__ __launch_bounds__(256) void gaussian_direct_kernel(GaussDirectArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const int R = a.radius;
    float hw = 0.0f;
    for (int dx = -R; dx <= R; ++dx)
        if (x + dx >= 0 && x + dx < a.W) hw += a.g[dx < 0 ? -dx : dx];
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, vw = 0.0f;
    for (int dy = -R; dy <= R; ++dy) {
        const int ty = y + dy;
        if (ty < 0 || ty >= a.H) continue;
        float hr = 0.0f, hg = 0.0f, hb = 0.0f;
        for (int dx = -R; dx <= R; ++dx) {
            const int tx = x + dx;
            if (tx < 0 || tx >= a.W) continue;
            const uchar4 c = a.in[(size_t)ty * a.W + tx];
            const float w = a.g[dx < 0 ? -dx : dx];
            hr = __builtin_fmaf(w, (float)c.x, hr); hg = __builtin_fmaf(w, (float)c.y, hg); hb = __builtin_fmaf(w, (float)c.z, hb);
        }
        const float w = a.g[dy < 0 ? -dy : dy];
        sr = __builtin_fmaf(w, hr, sr); sg = __builtin_fmaf(w, hg, sg); sb = __builtin_fmaf(w, hb, sb); vw += w;
    }
    const float sw = hw * vw;
    a.out[(size_t)y * a.W + x] = make_uchar4((unsigned char)(sr / sw), (unsigned char)(sg / sw), (unsigned char)(sb / sw), 0);
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
    if (p.type == RMD_FILTER_GAUSSIAN && p.radius > kGaussDirectMaxRadius)
        return fail(RMD_E_PARAM, "rmd_filter_tiled: GAUSSIAN radius %d > %d", p.radius, kGaussDirectMaxRadius);
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
        if (p.type == RMD_FILTER_GAUSSIAN && a.radius > kGaussMaxRadius) {          // beyond the separable kernel's LDS tile
            GaussDirectArgs gd;
            gd.in = a.in; gd.out = a.out; gd.W = W; gd.H = H; gd.radius = a.radius;
            for (int d = 0; d <= kGaussDirectMaxRadius; ++d) gd.g[d] = expf(-(float)(d * d) * a.inv2s_space);
            hipLaunchKernelGGL(gaussian_direct_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, stream, gd);
            RMD_LAUNCH_CHECK("gaussian_direct_kernel");
            continue;
        }
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
        if (a.radius == 2 && a.step <= kTileMaxStep && tuning_env("RMD_WEIGHTED_TILE", 1)) {  // CROSS radius 2, WAVELET levels 0..3
            const int rc = a.albedo && a.normal ? launch_weighted_tile<true, true>(a, stream)
                         : a.albedo             ? launch_weighted_tile<true, false>(a, stream)
                         : a.normal             ? launch_weighted_tile<false, true>(a, stream)
                                                : launch_weighted_tile<false, false>(a, stream);
            if (rc != RMD_OK) return rc;
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
