// svgf_atrous.hip — A pass: one edge-stopping a-trous iteration (SURVEY Appendix A.A).
//
// Reference footprint: the unused B3-spline taps (reference src/filter.cu:10), the 5x5 window
// (src/test.cu:75), tap order dx outer / dy inner (src/filter.cu:34-35), "skip OOB taps and
// renormalise" (src/filter.cu:38-39,49) and the per-level ping-pong (src/filter.cu:24-25).
// Dilation, the normal/depth/luminance weights and the variance channel are Appendix A.
//
// Two variants with IDENTICAL per-pixel arithmetic (same helpers, same summation order):
//
//  direct    one thread per pixel, taps read from global memory.  Any step; fallback + cross-check.
//
//  stream    the MI355X kernel.  At step S the image splits into S independent row lattices
//            (rows y = r mod S): a tap at +-S, +-2S rows stays in the pixel's lattice.  A
//            workgroup (4 wave64 = 256 threads) owns a column strip of one lattice inside one
//            band of rows and walks down it: NP groups of 256/NP threads, group p producing
//            lattice rows j+2p, j+2p+1 of the step (default NP = 2: 128 columns x 4 rows per
//            step; NP = 1: 256 columns x 2 rows).  A ring of 2NP+4 lattice rows (width
//            256/NP + 4S) lives in LDS, so every input row is fetched from L2/HBM once per
//            strip as 16-byte-per-lane coalesced segments and each of the 25 taps is a
//            conflict-free ds_read_b128 with an immediate offset.  Each thread produces two
//            vertically adjacent lattice pixels (A, B) per step: 30 tap fetches serve 50 weight
//            evaluations, and the 40 evaluations whose tap both pixels share run their cosine
//            and their five accumulations as PACKED f32 (v_pk_fma_f32 / v_pk_mul_f32 /
//            v_pk_add_f32 on the (A,B) register pair, the tap value broadcast through op_sel).
//            The next 2NP lattice rows are prefetched into registers while the current rows
//            are computed.  Workgroup ids are remapped so that
//            each XCD owns a contiguous run of (band, strip, lattice) work: neighbouring strips /
//            lattices, which share halo columns and the +-1 variance rows, hit the same L2.
//
// Measured on MI355X this pass is VALU-issue bound, not HBM bound (tools/microbench/valu_rate:
// one SIMD retires a wave64 f32 op every ~3.7-5 cycles, a transcendental every ~8.5, and a
// packed op every ~5.5-7): 25 taps x (3 FMA dot + log2 + exp2 + 2 edge terms + 5 accumulates)
// per pixel costs more issue slots than 48 B/px cost HBM time.  Hence the instruction diet:
//   - weights in the log2 domain, ONE v_log_f32 + ONE v_exp_f32 per tap:
//       w = exp2( log2 k + sigma_n*log2(clamp01(n_p.n_t)) - |dz|*log2e/(za*len+1e-8) - |dl|*log2e/l_den )
//   - luminance is computed once per STAGED pixel, not per tap: LDS holds (lum, r, g, var);
//     blue is accumulated through lum and recovered at the end, b = (L - .2126R - .7152G)/.0722
//   - normals are unit length by contract, so max(0, n.n) is applied as a [0,1] clamp.
#include "common.h"
#include <type_traits>

namespace rmd {

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef RMD_ATROUS_GROUP_ROWS
#define RMD_ATROUS_GROUP_ROWS 2   // window rows per scheduling group of the tap loop (1, 2, 3 or 6)
#endif

#ifndef RMD_PRIO_T1               // a wave's priority drops 3 -> 2 -> 1 -> 0 once T1/16, T2/16, T3/16 of its rows are done
#define RMD_PRIO_T1 8
#define RMD_PRIO_T2 12
#define RMD_PRIO_T3 14
#endif

struct AtrousArgs {
    Geom g;
    const float4* in; const float4* nd; float4* out;
    int row0, row1;
    int step;
    float sigma_n, sigma_z, sigma_l;
    // stream variant work decomposition
    int band_h, band_base, nstrips, nblocks, per_xcd;
    // pair kernel: strips that touch the left / right frame border run the slower per-lane-tested body and get
    // bands of half the height (xe_lo of them at the left, the rest of the non-interior ones at the right)
    int n_int, xe_lo, band_h_xe, total_int, int_per_xcd, xe_per_xcd;
    int n_hi, band_h_hi, nblocks_hi;   // stream kernel: the first n_hi/2 and last n_hi - n_hi/2 strips are cut into bands of band_h_hi (< band_h) rows
    int cus;       // CUs the launch may count on (rmd_svgf_params.atrous_cus or the whole device)
    int nt_out;    // store the outputs non-temporally (launches whose planes overflow the 256 MB Infinity Cache)
};

// log2 of the B3-spline taps {3/8, 1/4, 1/16} (reference src/filter.cu:10)
__device__ constexpr float kLogB3[3] = { -1.41503749927884381855f, -2.0f, -4.0f };
constexpr float kLumR = 0.2126f, kLumG = 0.7152f, kLumB = 0.0722f;

// ---- lane-generic helpers: the SAME operations on one pixel (float) or on the (A,B) pair (f2) --
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2    fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ f2    clamp01(f2 x) { return f2{ clamp01(x.x), clamp01(x.y) }; }
__device__ __forceinline__ float log2_(float x) { return fast_log2(x); }
__device__ __forceinline__ f2    log2_(f2 x) { return f2{ fast_log2(x.x), fast_log2(x.y) }; }
__device__ __forceinline__ float exp2_(float x) { return fast_exp2(x); }
__device__ __forceinline__ f2    exp2_(f2 x) { return f2{ fast_exp2(x.x), fast_exp2(x.y) }; }
__device__ __forceinline__ float bc(float v, float) { return v; }          // broadcast a tap scalar
__device__ __forceinline__ f2    bc(float v, f2) { return f2{ v, v }; }

// A staged tap: color plane holds (lum, r, g, var), nd plane (nx, ny, nz, z).
struct Tap { float4 c, n; };

__device__ __forceinline__ float4 to_lrgv(const float4 rgbv)
{
    return make_float4(lum3(rgbv.x, rgbv.y, rgbv.z), rgbv.x, rgbv.y, rgbv.w);
}

// Per-pixel constants.  T = float for one pixel, f2 for the (A,B) pair.
template <class T>
struct Center {
    T nx, ny, nz, z, lum, il;
};
struct CenterAux {          // per pixel, not packed (edge terms use abs/neg source modifiers)
    float iz[5];            // log2e/(za*len+1e-8) for len = 1, sqrt2, 2, sqrt5, 2*sqrt2
    bool zero;              // centre normal is (0,0,0)
};
template <class T>
struct Acc { T sw, sl, sr, sg, sv; };

__device__ __forceinline__ int len_class(int adx, int ady)
{
    const int m = adx * adx + ady * ady;            // 1,2,4,5,8
    return m == 1 ? 0 : m == 2 ? 1 : m == 4 ? 2 : m == 5 ? 3 : 4;
}

// A.A.1: 3x3 Gaussian {1/4, 1/8, 1/16} of the variance, interior form (weights sum to 1).
__device__ __forceinline__ float prefilter9(float ul, float l, float dl, float u, float c, float d, float ur, float r, float dr)
{
    float v = 0.0625f * ul;
    v = fma_(0.125f, l, v);  v = fma_(0.0625f, dl, v);
    v = fma_(0.125f, u, v);  v = fma_(0.25f, c, v);   v = fma_(0.125f, d, v);
    v = fma_(0.0625f, ur, v); v = fma_(0.125f, r, v); v = fma_(0.0625f, dr, v);
    return v;
}

__device__ __forceinline__ void make_center(const Tap& t, float var_c, float gz, float sigma_z, float sigma_l, float step,
                                            Center<float>& k, CenterAux& x)
{
    k.nx = t.n.x; k.ny = t.n.y; k.nz = t.n.z; k.z = t.n.w;
    k.lum = t.c.x;
    x.zero = is_zero3(t.n);
    // log2e / (a*len + 1e-8) is evaluated as 1 / (a*(len/log2e) + 1e-8/log2e): one fma + one v_rcp_f32
    constexpr float kInvLog2e = 1.0f / kLog2e, kEps = 1e-8f / kLog2e;
    const float vc = var_c > 0.0f ? var_c : 0.0f;
    k.il = fast_rcp(fma_(sigma_l * kInvLog2e, __builtin_amdgcn_sqrtf(vc), kEps));
    const float za = sigma_z * fmaxf(gz, 1e-8f) * step;
    x.iz[0] = fast_rcp(fma_(za, 1.0f * kInvLog2e, kEps));
    x.iz[1] = fast_rcp(fma_(za, 1.41421356237309504880f * kInvLog2e, kEps));
    x.iz[2] = fast_rcp(fma_(za, 2.0f * kInvLog2e, kEps));
    x.iz[3] = fast_rcp(fma_(za, 2.23606797749978969641f * kInvLog2e, kEps));
    x.iz[4] = fast_rcp(fma_(za, 2.82842712474619009760f * kInvLog2e, kEps));
}

__device__ __forceinline__ f2 pair_of(float a, float b)
{
    // opaque to the optimizer: without this the (A,B) pairs are assembled by spilling both
    // Center<float> structs to scratch and reloading overlapping <2 x float>s
    asm("" : "+v"(a), "+v"(b));
    return f2{ a, b };
}
__device__ __forceinline__ Center<f2> pack(const Center<float> a, const Center<float> b)
{
    Center<f2> k;
    k.nx = pair_of(a.nx, b.nx); k.ny = pair_of(a.ny, b.ny); k.nz = pair_of(a.nz, b.nz); k.z = pair_of(a.z, b.z);
    k.lum = pair_of(a.lum, b.lum); k.il = pair_of(a.il, b.il);
    return k;
}

// The [0,1] clamp of the cosine as the VOP3P clamp bit of its last packed fma: the compiler only folds
// a clamp into scalar v_fma_f32 and otherwise spends two v_max_f32 per pair.  `b` is the (z, w) half of
// a staged float4; same value as the generic form.  (Further hand-packed forms were tried and dropped:
// the two differences z_p - z_t, l_p - l_t as v_pk_add_f32 save two issue slots per pair but no time
// and cost 3 VGPRs, which moves the allocation from 152 to 160 registers and evicts the T wave that
// otherwise fits beside three a-trous waves; a packed variance accumulation through op_sel spills.)
__device__ __forceinline__ f2 pk_fma_lo_clamp(f2 a, f2 b, f2 c)        // clamp01(a * b.lo + c)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// clamp01(n_p . n_t): the same three operations for one pixel (float) and for the pair (f2)
template <class T>
__device__ __forceinline__ T tap_cosine(const Center<T>& k, const Tap& t)
{
    T d = k.nx * bc(t.n.x, T{});
    d = fma_(k.ny, bc(t.n.y, T{}), d);
    if constexpr (std::is_same<T, f2>::value) return pk_fma_lo_clamp(k.nz, f2{ t.n.z, t.n.w }, d);
    else return clamp01(fma_(k.nz, bc(t.n.z, T{}), d));
}

// One pixel's exponent, scalar on purpose: v_fma_f32 takes |.| and - as free source modifiers and
// e0 as a literal (packed f32 has neither, and a packed e0 would need a VGPR pair per tap):
//   e = e0 + sigma_n*log2(cos) - |z_p - z_t|*iz - |l_p - l_t|*il
template <bool ZERO_AWARE>
__device__ __forceinline__ float tap_exponent(float cosine, float e0, float sigma_n, float zp, float lp, float il,
                                              const CenterAux& x, const Tap& t, bool tap_zero, int adx, int ady)
{
    float e = fma_(sigma_n, log2_(cosine), e0);
    if (ZERO_AWARE) {   // Appendix A.A.2: both normals zero => w_n = 1, exactly one zero => 0
        if (x.zero) e = tap_zero ? e0 : kNegInf;
    }
    if (adx | ady) e = fma_(-fabsf(zp - t.n.w), x.iz[len_class(adx, ady)], e);
    return fma_(-fabsf(lp - t.c.x), il, e);
}

template <class T>
__device__ __forceinline__ void tap_accumulate(Acc<T>& s, T w, const Tap& t)
{
    s.sw += w;
    s.sl = fma_(w, bc(t.c.x, T{}), s.sl);
    s.sr = fma_(w, bc(t.c.y, T{}), s.sr);
    s.sg = fma_(w, bc(t.c.z, T{}), s.sg);
    s.sv = fma_(w * w, bc(t.c.w, T{}), s.sv);
}

// one pixel, one tap
template <bool ZERO_AWARE>
__device__ __forceinline__ void tap_single(Acc<float>& s, const Center<float>& k, const CenterAux& x, const Tap& t,
                                           float e0, int adx, int ady, float sigma_n)
{
    const float e = tap_exponent<ZERO_AWARE>(tap_cosine<float>(k, t), e0, sigma_n, k.z, k.lum, k.il, x, t,
                                             ZERO_AWARE && is_zero3(t.n), adx, ady);
    tap_accumulate<float>(s, exp2_(e), t);
}

// the (A,B) pair sharing one tap: adyA / adyB are the tap's |row offset| seen from A and from B.
// Cosine and accumulation are packed (v_pk_*), the exponents scalar; the empty asm keeps the
// optimizer from re-vectorising the exponents (it would gather iz[] pairs through scratch).
template <bool ZERO_AWARE>
__device__ __forceinline__ void tap_pair(Acc<f2>& s, const Center<f2>& k, const CenterAux& xa, const CenterAux& xb,
                                         const Tap& t, float e0A, float e0B, int adx, int adyA, int adyB, float sigma_n)
{
    const f2 c = tap_cosine<f2>(k, t);
    if constexpr (!ZERO_AWARE) {
        // Every VALU instruction of this loop costs a full issue slot, so what can be
        // packed is: sigma_n*log2(cos) + log2 k for both pixels (the two log2 k ride in an SGPR pair; sigma_n
        // is kept in a VGPR to leave the one scalar operand slot to them) and the two differences; only the
        // |.|-scaled terms stay scalar (packed f32 has no abs modifier).  Same operations as tap_exponent,
        // same bits.
        f2 e = fma_(f2{ sigma_n, sigma_n }, log2_(c), f2{ e0A, e0B });
        const f2 dz = k.z - f2{ t.n.w, t.n.w }, dl = k.lum - f2{ t.c.x, t.c.x };
        if (adx | adyA) e.x = fma_(-fabsf(dz.x), xa.iz[len_class(adx, adyA)], e.x);
        if (adx | adyB) e.y = fma_(-fabsf(dz.y), xb.iz[len_class(adx, adyB)], e.y);
        e.x = fma_(-fabsf(dl.x), k.il.x, e.x);
        e.y = fma_(-fabsf(dl.y), k.il.y, e.y);
        tap_accumulate<f2>(s, exp2_(e), t);
    } else {
        const bool tz = is_zero3(t.n);
        float ca = c.x, cb = c.y;
        asm("" : "+v"(ca), "+v"(cb));
        const float ea = tap_exponent<ZERO_AWARE>(ca, e0A, sigma_n, k.z.x, k.lum.x, k.il.x, xa, t, tz, adx, adyA);
        const float eb = tap_exponent<ZERO_AWARE>(cb, e0B, sigma_n, k.z.y, k.lum.y, k.il.y, xb, t, tz, adx, adyB);
        tap_accumulate<f2>(s, f2{ exp2_(ea), exp2_(eb) }, t);
    }
}

// A.A.3.  c = the centre in (lum, r, g, var) form.
__device__ __forceinline__ float4 finish(float sw, float sl, float sr, float sg, float sv, const float4 c)
{
    float L, R, G, V;
    if (sw < 1e-10f) { L = c.x; R = c.y; G = c.z; V = c.w; }      // pass-through
    else {
        const float inv = fast_rcp(sw);
        L = sl * inv; R = sr * inv; G = sg * inv; V = sv * inv * inv;
    }
    // blue is recovered from the luminance sum; the recovery amplifies the rounding of L by 1/0.0722, so
    // a true blue of 0 could come out as a tiny negative value and be fed back as history: clamped
    const float B = fmaxf(fma_(-kLumG, G, fma_(-kLumR, R, L)) * (1.0f / kLumB), 0.0f);
    return make_float4(R, G, B, V);
}

#ifdef RMD_ATROUS_TRACE
// tuning aid (tools/build_variant.sh trace -DRMD_ATROUS_TRACE): per-workgroup start / end time
// (100 MHz s_memrealtime), placement (XCC_ID, HW_ID) and code path of the LAST stream launch
__device__ unsigned long long g_atrous_trace[6 * 8192];
__device__ unsigned long long g_atrous_phase[8 * 8192];     // per workgroup (wave 0): cycles in load issue / compute / barrier 1 / store / barrier 2
#define RMD_PHASE(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tp; tp = tn_; }
#else
#define RMD_PHASE(i)
#endif

// ---------------------------------------------------------------------------------- direct
__global__ __launch_bounds__(256) void atrous_direct_kernel(AtrousArgs a)
{
    const Geom g = a.g;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= a.row1) return;
    const int s = a.step;
    const size_t i = pix_index(g, x, y);
    Tap ctr;
    ctr.c = to_lrgv(a.in[i]);
    ctr.n = a.nd[i];

    // A.A.1: 3x3 Gaussian prefilter of the variance, OOB skipped + renormalised
    float var_c;
    const bool okl = x - 1 >= 0, okr = x + 1 < g.W, oku = y - 1 >= 0, okd = y + 1 < g.H;
    auto var_at = [&](int tx, int ty) { return a.in[pix_index(g, tx, ty)].w; };
    if (okl && okr && oku && okd) {
        var_c = prefilter9(var_at(x - 1, y - 1), var_at(x - 1, y), var_at(x - 1, y + 1), var_at(x, y - 1), ctr.c.w,
                           var_at(x, y + 1), var_at(x + 1, y - 1), var_at(x + 1, y), var_at(x + 1, y + 1));
    } else {
        float gs = 0.25f, vs = 0.25f * ctr.c.w;
        if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x - 1, y - 1), vs); }
        if (okl)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x - 1, y), vs); }
        if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x - 1, y + 1), vs); }
        if (oku)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x, y - 1), vs); }
        if (okd)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x, y + 1), vs); }
        if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x + 1, y - 1), vs); }
        if (okr)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x + 1, y), vs); }
        if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x + 1, y + 1), vs); }
        var_c = vs / gs;
    }
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    const float gz = fabsf(a.nd[pix_index(g, x1, y)].w - ctr.n.w) + fabsf(a.nd[pix_index(g, x, y1)].w - ctr.n.w);
    Center<float> k;
    CenterAux aux;
    make_center(ctr, var_c, gz, a.sigma_z, a.sigma_l, (float)s, k, aux);

    // Same grouping as the stream kernel: a pixel whose lattice index floor(y/s) is even (role A)
    // sums its dy=-2 row apart from the other four rows, an odd one (role B) its dy=+2 row; each
    // group is summed dx outer / dy inner and the two groups are added at the end.
    const int lone_dy = ((y / s) & 1) ? 2 : -2;
    Acc<float> acc = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, lone = acc;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int tx = x + s * dx, ty = y + s * dy;
            if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) continue;
            const size_t ti = pix_index(g, tx, ty);
            Tap t;
            t.c = to_lrgv(a.in[ti]);
            t.n = a.nd[ti];
            const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            if (dy == lone_dy) tap_single<true>(lone, k, aux, t, kLogB3[adx] + kLogB3[ady], adx, ady, a.sigma_n);
            else               tap_single<true>(acc, k, aux, t, kLogB3[adx] + kLogB3[ady], adx, ady, a.sigma_n);
        }
    }
    a.out[i] = finish(lone.sw + acc.sw, lone.sl + acc.sl, lone.sr + acc.sr, lone.sg + acc.sg, lone.sv + acc.sv, ctr.c);
}

// ---------------------------------------------------------------------------------- stream
// NP = row pairs per workgroup.  The 256 threads are NP groups of CW = 256/NP columns; group p
// produces lattice rows j+2p (role A) and j+2p+1 (role B) of the step, so a step yields 2*NP rows
// of CW pixels from a ring of NR = 2*NP + 4 staged rows.  NP = 2 halves the LDS bytes per thread
// (8 rows for 4 output rows instead of 6 for 2), which is what lets 3 workgroups (12 waves) share
// a CU: one wave issues at most one instruction every ~6-7 cycles, so two waves per SIMD cannot
// saturate its VALU (tools/microbench/valu_rate).
template <int S, int NP>
struct StreamCfg {
    static constexpr int CW = 256 / NP;               // output columns per workgroup
    static constexpr int PW = CW + 4 * S;             // staged row width in pixels (halo 2S each side)
    static constexpr int NR = 2 * NP + 4;             // ring rows: j-2 .. j+2NP+1
    static constexpr int ADV = 2 * NP;                // lattice rows produced per step
    static constexpr int ROW_BYTES = PW * 16;
    static constexpr int PLANE_BYTES = NR * ROW_BYTES;
    static constexpr int VAR_OFF = 2 * PLANE_BYTES;   // 4*NP rows of CW+2 floats: variance of rows y-1 / y+1
    static constexpr int VAR_ROW = CW + 2;
    static constexpr int LDS_BYTES = VAR_OFF + 4 * NP * VAR_ROW * 4;
    static constexpr int TAIL = 16 * S * NP;          // float4 elements in the 4S-pixel row tails of one refill
    static constexpr int NT = (TAIL + 255) / 256;     // tail loads per thread
    static constexpr int WG_PER_CU = NP == 1 ? 2 : 3;      // LDS-limited (NP = 4: 3 as well, by registers)
};

__device__ __forceinline__ float4 lds_f4(const unsigned char* lds, int off) { return *reinterpret_cast<const float4*>(lds + off); }
__device__ __forceinline__ float  lds_f1(const unsigned char* lds, int off) { return *reinterpret_cast<const float*>(lds + off); }

// Lattice rows of this workgroup are y = ybase + j*S; ybase/S is even, so j even <=> pixel role A
// (global lattice index floor(y/S) even).  Outputs are wanted for j in [jlo, jhi); pairs (j, j+1)
// with j even are processed from j0 = jlo & ~1, a pixel whose row falls outside [jlo, jhi) is
// computed and dropped.  The pairing therefore depends only on (y, S), never on the row range or
// the band decomposition: outputs are bit-identical for every decomposition.
template <int S, int NP, bool EDGE>
__device__ __forceinline__ void atrous_stream_body(const AtrousArgs& a, unsigned char* lds, const int tid,
                                                   const int x0, const int ybase, const int jlo, const int jhi)
{
    using C = StreamCfg<S, NP>;
    const Geom g = a.g;
    const int col = tid % C::CW;                      // column inside the strip
    // row pair of this thread: the same for a whole wave (CW is a multiple of 64), said so explicitly
    // so that every row number and row address derived from it stays in SGPRs
    const int pr = __builtin_amdgcn_readfirstlane(tid / C::CW);
    const int x = x0 + col;
    const bool xin = !EDGE || x < g.W;
    const float* in_f = reinterpret_cast<const float*>(a.in);
    const float* nd_f = reinterpret_cast<const float*>(a.nd);
    float* var_lds = reinterpret_cast<float*>(lds + C::VAR_OFF);

    // ---- register prefetch state: two lattice rows (color, nd) + row tails, and for the next
    // output pair the variance of rows y-1 / y+1 (3x3 prefilter) and z of row y+1 (depth gradient)
    float4 pc[2], pn[2], pe[C::NT];
    float pvu[2], pvd[2], pzd[2], pvh = 0.0f;
    float zd_cur[2] = { 0.0f, 0.0f };
    float4 outA, outB;                                // results of the current step

    // Global addresses are formed as (wave-uniform row base) + (lane index): the uniform part
    // stays in SGPRs (global_load ... saddr), only the lane offset lives in a VGPR.
    auto row_base = [&](const int y, const int xs) -> long long {
        return (long long)(y - g.buf_row0) * (long long)g.W + (long long)xs;
    };
    // rows outside the frame are zero-filled and their taps masked; rows inside the frame but
    // outside the buffer can only be asked for by a dropped pixel and are zero-filled too
    auto row_in_buffer = [&](const int y) { return y >= max(g.buf_row0, 0) && y < min(g.buf_row0 + g.buf_rows, g.H); };
    auto slot_of = [&](const int j) { return (j + 4 * C::NR) % C::NR; };        // j >= -2

    // A refill brings ADV rows (jb .. jb+ADV-1).  Thread (pr, col) loads pixel `col` of rows
    // jb+2pr, jb+2pr+1 of both planes; the 4S-pixel tails of the ADV rows x 2 planes (TAIL float4
    // in all) are spread over the threads: tail element e -> row (e/4S)>>1, plane (e/4S)&1,
    // column CW + e%4S.
    // (NROWS < ADV: only the first NROWS rows of the refill are fetched -- the prologue of the NP = 4 ring, whose
    // 12 rows are one and a half refills; the rows beyond may lie outside the buffer)
    auto load_rows_n = [&](const int jb, auto nrows_c) {
        constexpr int NROWS = decltype(nrows_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = ybase + (jb + 2 * pr + i) * S;
            bool act = !EDGE || (row_in_buffer(y) && x - 2 * S >= 0 && x - 2 * S < g.W);
            if (NROWS < C::ADV) act = act && 2 * pr + i < NROWS;
            float4 vc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vn = vc;
            if (act) {
                const long long o = row_base(y, x0 - 2 * S);
                vc = (a.in + o)[col];
                vn = (a.nd + o)[col];
            }
            pc[i] = vc;
            pn[i] = vn;
        }
#pragma unroll
        for (int q = 0; q < C::NT; ++q) {
            const int e = tid + q * 256;
            const int sel = e / (4 * S);
            const int y = ybase + (jb + (sel >> 1)) * S;
            const int gx = x0 - 2 * S + C::CW + e % (4 * S);
            bool act = e < C::TAIL;
            if (NROWS < C::ADV) act = act && (sel >> 1) < NROWS;
            if (EDGE) act = act && row_in_buffer(y) && gx >= 0 && gx < g.W;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (act) {
                const float4* plane = (sel & 1) ? a.nd : a.in;
                v = plane[(size_t)(y - g.buf_row0) * (size_t)g.W + (size_t)gx];
            }
            pe[q] = v;
        }
    };
    auto load_rows = [&](const int jb) { load_rows_n(jb, std::integral_constant<int, C::ADV>{}); };
    // colour is staged as (lum, r, g, var): luminance once per staged pixel instead of once per tap
    // (nrows < ADV: only the first nrows rows of the refill are stored -- the prologue of the NP = 4 ring,
    // whose 12 rows are one and a half refills)
    auto store_rows = [&](const int jb, const int nrows = C::ADV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (2 * pr + i >= nrows) continue;
            const int off = slot_of(jb + 2 * pr + i) * C::ROW_BYTES + col * 16;
            *reinterpret_cast<float4*>(lds + off) = to_lrgv(pc[i]);
            *reinterpret_cast<float4*>(lds + C::PLANE_BYTES + off) = pn[i];
        }
#pragma unroll
        for (int q = 0; q < C::NT; ++q) {
            const int e = tid + q * 256;
            if (e < C::TAIL) {
                const int sel = e / (4 * S);
                if ((sel >> 1) >= nrows) continue;
                const bool is_nd = (sel & 1) != 0;
                const int off = (is_nd ? C::PLANE_BYTES : 0) + slot_of(jb + (sel >> 1)) * C::ROW_BYTES + (C::CW + e % (4 * S)) * 16;
                *reinterpret_cast<float4*>(lds + off) = is_nd ? pe[q] : to_lrgv(pe[q]);
            }
        }
    };
    // aux rows of the outputs of step jo: thread (pr, col) serves its own two pixels
    auto load_aux = [&](const int jo) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int jj = jo + 2 * pr + i;
            const int y = ybase + jj * S;
            const bool rowok = jj >= jlo && jj < jhi && xin;
            float vu = 0.0f, vd = 0.0f, zd = 0.0f;
            if (rowok) {
                if (!EDGE || y - 1 >= 0) vu = (in_f + row_base(y - 1, x0) * 4 + 3)[col * 4];
                if (!EDGE || y + 1 < g.H) vd = (in_f + row_base(y + 1, x0) * 4 + 3)[col * 4];
                const int yz = EDGE ? min(y + 1, g.H - 1) : y + 1;
                zd = (nd_f + row_base(yz, x0) * 4 + 3)[col * 4];
            }
            pvu[i] = vu; pvd[i] = vd; pzd[i] = zd;
        }
        pvh = 0.0f;
        if (tid < 8 * NP) {   // halo columns x0-1 and x0+CW of the 4*NP variance rows
            const int ii = tid >> 2, ud = (tid >> 1) & 1, side = tid & 1;
            const int jj = jo + ii;
            const int yy = ybase + jj * S + (ud ? 1 : -1);
            const int xx = side ? x0 + C::CW : x0 - 1;
            bool ok = jj >= jlo && jj < jhi;
            if (EDGE) ok = ok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            if (ok) pvh = in_f[((size_t)(yy - g.buf_row0) * (size_t)g.W + (size_t)xx) * 4 + 3];
        }
    };
    auto store_aux = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            var_lds[((2 * pr + i) * 2 + 0) * C::VAR_ROW + col + 1] = pvu[i];
            var_lds[((2 * pr + i) * 2 + 1) * C::VAR_ROW + col + 1] = pvd[i];
            zd_cur[i] = pzd[i];
        }
        if (tid < 8 * NP) {
            const int ii = tid >> 2, ud = (tid >> 1) & 1, side = tid & 1;
            var_lds[(ii * 2 + ud) * C::VAR_ROW + (side ? C::CW + 1 : 0)] = pvh;
        }
    };

    // The same fetches as load_rows + load_aux, ONE vector-memory instruction per call, for the interior form:
    // issued one per tap group inside the tap loop instead of together at the top of a step, where 4 waves x ~13
    // of them queue up in the CU's texture address unit and every wave stalls at issue (tools/atrous_trace.py:
    // 16-23 % of a step).  Unconditional, at clamped rows (the last step fetches rows nobody uses; rows of
    // dropped outputs read something harmless), so no branch joins inside the loop.
    constexpr int kPieces = 4 + C::NT + 7;
    const int blo_i = max(g.buf_row0, 0), bhi_i = min(g.buf_row0 + g.buf_rows, g.H);
    auto clamp_row = [&](const int y) { return min(max(y, blo_i), bhi_i - 1); };
    auto prefetch_piece = [&](const int i, const int jb, const int jo) {
        if (i < 4) {
            const int r = i >> 1;
            const long long o = row_base(clamp_row(ybase + (jb + 2 * pr + r) * S), x0 - 2 * S);
            if (i & 1) pn[r] = (a.nd + o)[col]; else pc[r] = (a.in + o)[col];
        } else if (i < 4 + C::NT) {
            const int q = i - 4;
            const int e = min(tid + q * 256, C::TAIL - 1);
            const int sel = e / (4 * S);
            const int y = clamp_row(ybase + (jb + (sel >> 1)) * S);
            const int gx = x0 - 2 * S + C::CW + e % (4 * S);
            const float4* plane = (sel & 1) ? a.nd : a.in;
            pe[q] = plane[(size_t)(y - g.buf_row0) * (size_t)g.W + (size_t)gx];
        } else if (i < 4 + C::NT + 6) {
            const int k = i - 4 - C::NT, r = k / 3, which = k % 3;
            const int y = clamp_row(ybase + (jo + 2 * pr + r) * S);
            if (which == 0)      pvu[r] = (in_f + row_base(clamp_row(y - 1), x0) * 4 + 3)[col * 4];
            else if (which == 1) pvd[r] = (in_f + row_base(clamp_row(y + 1), x0) * 4 + 3)[col * 4];
            else                 pzd[r] = (nd_f + row_base(clamp_row(y + 1), x0) * 4 + 3)[col * 4];
        } else {
            // halo columns x0-1 and x0+CW of the 4*NP variance rows: lanes 0 .. 8NP-1 keep theirs
            const int t8 = tid & (8 * NP - 1);
            const int ii = t8 >> 2, ud = (t8 >> 1) & 1, side = t8 & 1;
            const int yy = clamp_row(ybase + (jo + ii) * S + (ud ? 1 : -1));
            const int xx = side ? x0 + C::CW : x0 - 1;
            pvh = in_f[((size_t)(yy - g.buf_row0) * (size_t)g.W + (size_t)xx) * 4 + 3];
        }
    };

    // ---- per-pixel setup (A.A.1 prefilter, depth gradient) from the staged data
    auto setup = [&](const int i, const int y, const int rb_center, const Tap& t, Center<float>& k, CenterAux& aux) {
        const int ccol = rb_center + 2 * S * 16;      // byte offset of the centre pixel in the color plane
        const float* vu = var_lds + ((2 * pr + i) * 2 + 0) * C::VAR_ROW + col;     // [0..2] = x-1, x, x+1 of row y-1
        const float* vd = var_lds + ((2 * pr + i) * 2 + 1) * C::VAR_ROW + col;
        const float v_l = lds_f1(lds, ccol - 16 + 12), v_r = lds_f1(lds, ccol + 16 + 12);
        float var_c;
        bool interior = true;
        bool okl = true, okr = true, oku = true, okd = true;
        if (EDGE) {
            okl = x - 1 >= 0; okr = x + 1 < g.W; oku = y - 1 >= 0; okd = y + 1 < g.H;
            interior = okl && okr && oku && okd;
        }
        if (interior) {
            var_c = prefilter9(vu[0], v_l, vd[0], vu[1], t.c.w, vd[1], vu[2], v_r, vd[2]);
        } else {
            float gs = 0.25f, vs = 0.25f * t.c.w;
            if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, vu[0], vs); }
            if (okl)        { gs += 0.125f;  vs = fma_(0.125f, v_l, vs); }
            if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, vd[0], vs); }
            if (oku)        { gs += 0.125f;  vs = fma_(0.125f, vu[1], vs); }
            if (okd)        { gs += 0.125f;  vs = fma_(0.125f, vd[1], vs); }
            if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, vu[2], vs); }
            if (okr)        { gs += 0.125f;  vs = fma_(0.125f, v_r, vs); }
            if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, vd[2], vs); }
            var_c = vs / gs;
        }
        float zr = lds_f1(lds, C::PLANE_BYTES + ccol + 16 + 12);
        if (EDGE && !okr) zr = t.n.w;
        const float gz = fabsf(zr - t.n.w) + fabsf(zd_cur[i] - t.n.w);
        make_center(t, var_c, gz, a.sigma_z, a.sigma_l, (float)S, k, aux);
    };

    // ---- one step: this thread's outputs are lattice rows jw (A) and jw+1 (B), jw = j + 2*pr
    auto compute = [&](const int j) {
        const int jw = j + 2 * pr;
        int rb[6];
        {
            int slot = slot_of(jw - 2);
#pragma unroll
            for (int tr = 0; tr < 6; ++tr) {
                rb[tr] = slot * C::ROW_BYTES + col * 16;
                slot = slot == C::NR - 1 ? 0 : slot + 1;
            }
        }
        const int yA = ybase + jw * S, yB = yA + S;
        Tap cA, cB;
        cA.c = lds_f4(lds, rb[2] + 2 * S * 16); cA.n = lds_f4(lds, C::PLANE_BYTES + rb[2] + 2 * S * 16);
        cB.c = lds_f4(lds, rb[3] + 2 * S * 16); cB.n = lds_f4(lds, C::PLANE_BYTES + rb[3] + 2 * S * 16);
        Center<float> kA, kB;
        CenterAux xA, xB;
        setup(0, yA, rb[2], cA, kA, xA);
        setup(1, yB, rb[3], cB, kB, xB);
        const Center<f2> kAB = pack(kA, kB);
        kA = Center<float>{ kAB.nx.x, kAB.ny.x, kAB.nz.x, kAB.z.x, kAB.lum.x, kAB.il.x };   // lanes, not copies
        kB = Center<float>{ kAB.nx.y, kAB.ny.y, kAB.nz.y, kAB.z.y, kAB.lum.y, kAB.il.y };
        Acc<f2> sAB = { f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 } };
        Acc<float> sA = { 0, 0, 0, 0, 0 }, sB = sA;      // rows only one of the two pixels taps (tr = 0 / tr = 5)

        bool rowv[6];
#pragma unroll
        for (int tr = 0; tr < 6; ++tr) {
            const int yy = yA + (tr - 2) * S;
            rowv[tr] = !EDGE || (yy >= 0 && yy < g.H);
        }
        // a wave takes the cheaper path when none of its centres has a zero normal
        const bool any_zero = __builtin_amdgcn_ballot_w64(xA.zero || xB.zero) != 0ull;
        float sn;                                                      // sigma_n in a VGPR (see tap_pair)
        asm volatile("v_mov_b32 %0, %1" : "=v"(sn) : "s"(a.sigma_n));

        // The 30 taps of a step are walked in groups of GR window rows (dx outer, rows inner):
        // the ds_read_b128 of group g+1 are issued before group g is weighted, and a scheduling fence
        // per group keeps the compiler from hoisting every read of the step to the top (that needs
        // more VGPRs than 3 waves per SIMD leave: 168).
        constexpr int GR = RMD_ATROUS_GROUP_ROWS, NG = 5 * (6 / GR);
        auto load_grp = [&](const int grp, Tap (&t)[GR]) {
            const int dxi = grp / (6 / GR), tr0 = (grp % (6 / GR)) * GR;
#pragma unroll
            for (int q = 0; q < GR; ++q) {
                const int off = rb[tr0 + q] + (2 * S + (dxi - 2) * S) * 16;
                t[q].c = lds_f4(lds, off);
                t[q].n = lds_f4(lds, C::PLANE_BYTES + off);
            }
        };
        auto taps = [&](auto zero_aware) {
            constexpr bool ZA = decltype(zero_aware)::value;
            // per-output order stays dx outer / dy inner: A sees rows 0..4 as dy=-2..2, B rows 1..5
            auto weigh_grp = [&](const int grp, const Tap (&t)[GR]) {
                const int dxi = grp / (6 / GR), tr0 = (grp % (6 / GR)) * GR;
                const int dx = dxi - 2;
                const int adx = dx < 0 ? -dx : dx;
                const bool colv = !EDGE || (x + dx * S >= 0 && x + dx * S < g.W);
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    const int tr = tr0 + q;
                    // Taps outside the frame are staged as zeros.  A zero tap normal gives
                    // log2(0) = -inf, i.e. weight exactly 0, whenever the centre normal is non-zero,
                    // so only the zero-aware path needs an explicit mask.
                    const bool valid = !EDGE || !ZA || (colv && rowv[tr]);
                    const int dyA = tr - 2, dyB = tr - 3;
                    const int adyA = dyA < 0 ? -dyA : dyA, adyB = dyB < 0 ? -dyB : dyB;
                    // log2 k of the tap, or -inf for a tap outside the frame (w becomes exactly 0)
                    const float e0A = valid ? kLogB3[adx] + kLogB3[adyA] : kNegInf;
                    const float e0B = valid ? kLogB3[adx] + kLogB3[adyB] : kNegInf;
                    if (tr == 0)      tap_single<ZA>(sA, kA, xA, t[q], e0A, adx, adyA, sn);
                    else if (tr == 5) tap_single<ZA>(sB, kB, xB, t[q], e0B, adx, adyB, sn);
                    else              tap_pair<ZA>(sAB, kAB, xA, xB, t[q], e0A, e0B, adx, adyA, adyB, sn);
                }
            };
            Tap t0[GR], t1[GR];
            load_grp(0, t0);
            static_assert(kPieces <= NG || EDGE, "one refill instruction per tap group");
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                if (!EDGE && grp < kPieces) prefetch_piece(grp, j - 2 + C::NR, j + C::ADV);
                if (grp & 1) { if (grp < NG - 1) load_grp(grp + 1, t0); weigh_grp(grp, t1); }
                else         { if (grp < NG - 1) load_grp(grp + 1, t1); weigh_grp(grp, t0); }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (any_zero) taps(std::true_type{}); else taps(std::false_type{});

        // Per-pixel sums = (row only this pixel taps: dy=-2 for role A, dy=+2 for role B) + (the four
        // rows shared with its partner), each summed dx outer / dy inner.  The direct kernel groups
        // its 25 taps the same way, so the two variants stay bit-identical.
        outA = finish(sA.sw + sAB.sw.x, sA.sl + sAB.sl.x, sA.sr + sAB.sr.x, sA.sg + sAB.sg.x, sA.sv + sAB.sv.x, cA.c);
        outB = finish(sB.sw + sAB.sw.y, sB.sl + sAB.sl.y, sB.sr + sAB.sr.y, sB.sg + sAB.sg.y, sB.sv + sAB.sv.y, cB.c);
    };
    // The two output pixels of step j are written AFTER the ring refill of the step: the refill has to
    // wait for the prefetch loads (vmcnt), and stores issued before it would be waited for as well.
    auto write_out = [&](const int j) {
        const int jw = j + 2 * pr;
        const int yA = ybase + jw * S;
        if (xin) {
            // Non-temporal stores: the 133 MB a 4K launch writes are the expensive third of its traffic (the
            // skeleton without them runs at 5.4 TB/s, with them at 4.85), and nothing reads an output row
            // again before the next launch.  Launch for launch 131-145 -> 121-131 us, inside a frame 0.960 ->
            // 0.948 ms (the next iteration then fetches all of its input from HBM).  Non-temporal LOADS of the
            // colour plane lose: its halo columns are shared with the neighbour strip through L2.
            // Only where the launch's three planes overflow the Infinity Cache anyway: a smaller launch (one rank's
            // strip of an 8K frame: 3 x 84 MB) finds its input still cached if the previous one stored normally.
            typedef float f4n __attribute__((ext_vector_type(4)));
            if (a.nt_out) {
                if (jw >= jlo && jw < jhi) __builtin_nontemporal_store(f4n{ outA.x, outA.y, outA.z, outA.w }, reinterpret_cast<f4n*>(a.out + row_base(yA, x0)) + col);
                if (jw + 1 >= jlo && jw + 1 < jhi) __builtin_nontemporal_store(f4n{ outB.x, outB.y, outB.z, outB.w }, reinterpret_cast<f4n*>(a.out + row_base(yA + S, x0)) + col);
            } else {
                if (jw >= jlo && jw < jhi) (a.out + row_base(yA, x0))[col] = outA;
                if (jw + 1 >= jlo && jw + 1 < jhi) (a.out + row_base(yA + S, x0))[col] = outB;
            }
        }
    };

    // ---- prologue: ring rows j0-2 .. j0-2+NR-1 (NR/ADV refills) and the aux rows of the first step
    const int j0 = jlo & ~1;
    constexpr int kFull = C::NR / C::ADV, kRest = C::NR % C::ADV;
#pragma unroll
    for (int q = 0; q < kFull; ++q) {
        load_rows(j0 - 2 + q * C::ADV);
        if (kRest == 0 && q == kFull - 1) load_aux(j0);
        store_rows(j0 - 2 + q * C::ADV);
    }
    if (kRest != 0) {
        load_rows_n(j0 - 2 + kFull * C::ADV, std::integral_constant<int, (kRest ? kRest : C::ADV)>{});
        load_aux(j0);
        store_rows(j0 - 2 + kFull * C::ADV, kRest);
    }
    store_aux();
    __syncthreads();

    // The SIMD arbiter issues oldest-wave-first.  With three VALU-bound workgroups on a CU that
    // lets the oldest finish at ~half the kernel time and the youngest run its second half with
    // one wave per SIMD, which cannot fill the VALU (tools/atrous_trace.py: end times spread over
    // 80..155 us).  Lowering a wave's priority as ITS work completes makes the three workgroups
    // advance together, so the CU keeps three waves per SIMD until nearly the end.
    const int span = jhi - j0;
    const int jq1 = j0 + span * RMD_PRIO_T1 / 16, jq2 = j0 + span * RMD_PRIO_T2 / 16, jq3 = j0 + span * RMD_PRIO_T3 / 16;
    __builtin_amdgcn_s_setprio(3);
#ifdef RMD_ATROUS_TRACE
    unsigned long long ph[5] = { 0, 0, 0, 0, 0 }, tp = __builtin_amdgcn_s_memtime();
#endif
    for (int j = j0; j < jhi; j += C::ADV) {
        if (j >= jq3)      __builtin_amdgcn_s_setprio(0);
        else if (j >= jq2) __builtin_amdgcn_s_setprio(1);
        else if (j >= jq1) __builtin_amdgcn_s_setprio(2);
        const bool more = j + C::ADV < jhi;
        if (EDGE && more) { load_rows(j - 2 + C::NR); load_aux(j + C::ADV); }     // in flight during compute (interior form: inside compute)
        RMD_PHASE(0)
        compute(j);
        RMD_PHASE(1)
        if (!more) { write_out(j); break; }
        __syncthreads();                                     // every wave is done reading the ADV oldest rows
        RMD_PHASE(2)
        store_rows(j - 2 + C::NR); store_aux();
        write_out(j);
        RMD_PHASE(3)
        __syncthreads();
        RMD_PHASE(4)
    }
#ifdef RMD_ATROUS_TRACE
    if (tid == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 5; ++i) g_atrous_phase[8 * blockIdx.x + i] = ph[i];
#endif
}


template <int S, int NP>
__global__ __launch_bounds__(256, (NP == 1 ? 2 : 3)) void atrous_stream_kernel(AtrousArgs a)
{
#ifdef RMD_ATROUS_TRACE
    const unsigned long long trace_t0 = wall_clock64(), trace_c0 = __builtin_amdgcn_s_memtime();
#endif
    using C = StreamCfg<S, NP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char atrous_lds[];
    const int tid = threadIdx.x;
    // XCD-aware remap: workgroups pid, pid+8, ... share an XCD (round-robin dispatch), give each
    // XCD one contiguous run of logical work so halo columns / variance rows are shared in its L2.
    const int pid = blockIdx.x;
    const int L = (pid & (kXcds - 1)) * a.per_xcd + (pid >> 3);
    if (L >= a.nblocks) return;
    // Two groups of strips: n_hi strips (the outermost ones: their frame-edge body is the slower one) are cut into
    // one band more than the others, so that the workgroup count lands on the resident slots (plan_stream).
    int strip, band, bh;
    const int r = L % S;
    if (L < a.nblocks_hi) {
        const int t = L / S, e = t % a.n_hi;
        strip = e < a.n_hi / 2 ? e : a.nstrips - (a.n_hi - e); band = t / a.n_hi; bh = a.band_h_hi;
    } else {
        const int t = (L - a.nblocks_hi) / S, n_lo = a.nstrips - a.n_hi;
        strip = a.n_hi / 2 + t % n_lo; band = t / n_lo; bh = a.band_h;
    }
    const int x0 = strip * C::CW;
    // bands start at band_base + k*bh; band_base is row0 rounded down to a multiple of 2S, which is
    // all the (A,B) pairing needs (global lattice index floor(y/S) even at the top of a band)
    const int yb = a.band_base + band * bh;
    const int lo = max(yb, a.row0), hi = min(yb + bh, a.row1);
    const int ybase = yb + r;
    const int jlo = lo > ybase ? (lo - ybase + S - 1) / S : 0;
    const int jhi = hi > ybase ? (hi - ybase + S - 1) / S : 0;      // exclusive
    if (jlo >= jhi) return;
    // lattice rows touched: j0-2 .. j0 + nsteps*ADV + 1
    const int j0 = jlo & ~1;
    const int nsteps = (jhi - j0 + C::ADV - 1) / C::ADV;
    const int ytop = ybase + (j0 - 2) * S, ybot = ybase + (j0 + nsteps * C::ADV + 1) * S;
    const int blo = max(a.g.buf_row0, 0), bhi = min(a.g.buf_row0 + a.g.buf_rows, a.g.H);
    const bool edge = (x0 - 2 * S < 0) || (x0 + C::CW + 2 * S > a.g.W) || ytop < blo || ybot >= bhi;
    if (edge) atrous_stream_body<S, NP, true>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
    else      atrous_stream_body<S, NP, false>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
#ifdef RMD_ATROUS_TRACE
    __syncthreads();
    if (tid == 0 && pid < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_atrous_trace[6 * pid + 0] = trace_t0;
        g_atrous_trace[6 * pid + 1] = wall_clock64();
        g_atrous_trace[6 * pid + 2] = ((unsigned long long)xcc << 32) | hw;
        g_atrous_trace[6 * pid + 3] = ((unsigned long long)L << 8) | (edge ? 1u : 0u);
        g_atrous_trace[6 * pid + 4] = __builtin_amdgcn_s_memtime() - trace_c0;      // shader-clock cycles of this workgroup
        g_atrous_trace[6 * pid + 5] = (unsigned long long)(jhi - jlo);              // lattice rows it produced
    }
#endif
}

#ifdef RMD_ATROUS_TRACE
extern "C" int rmd_debug_atrous_trace(unsigned long long* host_dst, int workgroups)
{
    RMD_HIP(hipDeviceSynchronize());
    RMD_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_atrous_trace), sizeof(unsigned long long) * 6 * (size_t)workgroups));
    if (const char* e = getenv("RMD_TRACE_PHASES")) {        // append the phase sums of the pair kernel behind the 6-word records
        (void)e;
        RMD_HIP(hipMemcpyFromSymbol(host_dst + 6 * (size_t)workgroups, HIP_SYMBOL(g_atrous_phase), sizeof(unsigned long long) * 8 * (size_t)workgroups));
    }
    void* dev = nullptr;                                   // cleared on read: the next read sees one launch only
    RMD_HIP(hipGetSymbolAddress(&dev, HIP_SYMBOL(g_atrous_trace)));
    RMD_HIP(hipMemset(dev, 0, sizeof(g_atrous_trace)));
    return RMD_OK;
}
#endif



// =================================================================================== pair kernels
// Second formulation (round 2).  tools/microbench/valu_issue.hip + valu_pairing.hip show that in a
// mixed instruction stream EVERY VALU instruction of a wave64 costs one ~4.2-cycle issue slot of its
// SIMD (transcendentals 8.2): plain v_fma/v_mul/v_add only run two-per-slot when two waves issue
// nothing else, which a real kernel never does.  So the cost of a step is its instruction COUNT, and a
// packed v_pk_*_f32 (two pixels per slot) halves every operation it can express.  Hence:
//   - a lane owns two HORIZONTALLY adjacent output pixels (x, x+1), x even.  Both see every tap at
//     the same offset (dx, dy): same B3 weight, same length class, no tap rows that only one of the
//     two uses -- all 24 off-centre taps run the same 16 + 4 instruction body;
//   - LDS holds the staged rows as 8 planes (l, r, g, var, nx, ny, nz, z) of floats, so the values of
//     the taps of the two pixels (columns c, c+1) are ONE aligned ds_read_b64 = one register pair;
//   - cosine (3), sigma_n*log2(cos)+log2 k (1), the two differences (2) and the five accumulations
//     (6) are packed; only the two |.|-modified fmas per pixel (packed f32 has no abs) stay scalar;
//   - the centre tap is the constant k(0,0) = 9/64 (w_n = 1, dz = dl = 0): no transcendental.
// Per-pixel order of summation is plain dx outer / dy inner (the oracle's), identical in the direct
// variant below, so the two stay bit-identical for every decomposition.

template <class T> struct TapT { T l, r, g, v, nx, ny, nz, z; };
template <class T> struct CenterT { T nx, ny, nz, z, l, il; T iz[5]; T zc; };   // zc = 1 if the centre normal is (0,0,0), else 0
template <class T> struct AccT { T sw, sl, sr, sg, sv; };

__device__ __forceinline__ float splat(float v, float) { return v; }
__device__ __forceinline__ f2    splat(float v, f2) { return f2{ v, v }; }
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) { return clamp01(fma_(a, b, c)); }
__device__ __forceinline__ f2    fma_clamp01(f2 a, f2 b, f2 c)
{
    f2 r;   // the compiler folds a clamp only into scalar fmas
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// e = c - |dz|*iz - |dl|*il, scalar per pixel on purpose (abs/neg are free source modifiers of v_fma_f32)
__device__ __forceinline__ float edge_terms(float c, float dz, float iz, float dl, float il)
{
    return fma_(-fabsf(dl), il, fma_(-fabsf(dz), iz, c));
}
__device__ __forceinline__ f2 edge_terms(f2 c, f2 dz, f2 iz, f2 dl, f2 il)
{
    return f2{ edge_terms(c.x, dz.x, iz.x, dl.x, il.x), edge_terms(c.y, dz.y, iz.y, dl.y, il.y) };
}

// One off-centre tap for one pixel (T = float) or for the pixel pair (T = f2).  ZA (zero-aware): the
// wave holds a centre whose normal is (0,0,0) (Appendix A.A.2: both zero => w_n = 1, exactly one zero
// => 0).  With ft = 1 - n_t.n_t (1 for a zero tap normal, ~0 for a unit one) the cosine becomes
// clamp01(n_p.n_t + zc*ft): unchanged bits for zc = 0, ft for a zero centre.
template <class T, bool ZA>
__device__ __forceinline__ void tap_eval(AccT<T>& s, const CenterT<T>& k, const TapT<T>& t, const T e0, const int cls, const T sigma_n)
{
    T d = k.nx * t.nx;
    d = fma_(k.ny, t.ny, d);
    T cosine;
    if (ZA) {
        T ft = fma_(-t.nx, t.nx, splat(1.0f, T{}));
        ft = fma_(-t.ny, t.ny, ft);
        ft = fma_(-t.nz, t.nz, ft);
        d = fma_(k.nz, t.nz, d);
        cosine = fma_clamp01(k.zc, ft, d);
    } else {
        cosine = fma_clamp01(k.nz, t.nz, d);
    }
    const T c = fma_(sigma_n, log2_(cosine), e0);
    const T e = edge_terms(c, k.z - t.z, k.iz[cls], k.l - t.l, k.il);
    const T w = exp2_(e);
    s.sw += w;
    s.sl = fma_(w, t.l, s.sl);
    s.sr = fma_(w, t.r, s.sr);
    s.sg = fma_(w, t.g, s.sg);
    s.sv = fma_(w * w, t.v, s.sv);
}
// the centre tap: w = k(0,0) = 9/64 exactly
template <class T>
__device__ __forceinline__ void tap_center(AccT<T>& s, const T l, const T r, const T g, const T v)
{
    const T w = splat(0.140625f, T{});
    s.sw += w;
    s.sl = fma_(w, l, s.sl);
    s.sr = fma_(w, r, s.sr);
    s.sg = fma_(w, g, s.sg);
    s.sv = fma_(w * w, v, s.sv);
}

template <class T> __device__ __forceinline__ T max0(T x);
template <> __device__ __forceinline__ float max0<float>(float x) { return x > 0.0f ? x : 0.0f; }
template <> __device__ __forceinline__ f2 max0<f2>(f2 x) { return f2{ max0<float>(x.x), max0<float>(x.y) }; }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ f2    sqrt_(f2 x) { return f2{ sqrt_(x.x), sqrt_(x.y) }; }
__device__ __forceinline__ float rcp_(float x) { return fast_rcp(x); }
__device__ __forceinline__ f2    rcp_(f2 x) { return f2{ fast_rcp(x.x), fast_rcp(x.y) }; }
__device__ __forceinline__ float maxf_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2    maxf_(f2 a, float b) { return f2{ fmaxf(a.x, b), fmaxf(a.y, b) }; }

// per-pixel constants from the centre values, the prefiltered variance and the depth gradient
// (same operations as make_center above)
template <class T>
__device__ __forceinline__ void center_consts(CenterT<T>& k, const T var_c, const T gz, const float sigma_z, const float sigma_l, const float step)
{
    constexpr float kInvLog2e = 1.0f / kLog2e, kEps = 1e-8f / kLog2e;
    k.il = rcp_(fma_(splat(sigma_l * kInvLog2e, T{}), sqrt_(max0<T>(var_c)), splat(kEps, T{})));
    const T za = maxf_(gz, 1e-8f) * splat(sigma_z, T{}) * splat(step, T{});
    k.iz[0] = rcp_(fma_(za, splat(1.0f * kInvLog2e, T{}), splat(kEps, T{})));
    k.iz[1] = rcp_(fma_(za, splat(1.41421356237309504880f * kInvLog2e, T{}), splat(kEps, T{})));
    k.iz[2] = rcp_(fma_(za, splat(2.0f * kInvLog2e, T{}), splat(kEps, T{})));
    k.iz[3] = rcp_(fma_(za, splat(2.23606797749978969641f * kInvLog2e, T{}), splat(kEps, T{})));
    k.iz[4] = rcp_(fma_(za, splat(2.82842712474619009760f * kInvLog2e, T{}), splat(kEps, T{})));
}

// A.A.3 for one pixel; c = (l, r, g, var) of the centre.  Blue is recovered from the luminance sum and
// clamped at 0 (the recovery amplifies the rounding of L by 1/0.0722; a true blue of 0 must not turn
// into a small negative history value).
__device__ __forceinline__ float4 finish2(float sw, float sl, float sr, float sg, float sv, float cl, float cr, float cg, float cv)
{
    float L, R, G, V;
    if (sw < 1e-10f) { L = cl; R = cr; G = cg; V = cv; }
    else {
        const float inv = fast_rcp(sw);
        L = sl * inv; R = sr * inv; G = sg * inv; V = sv * inv * inv;
    }
    const float B = fmaxf(fma_(-kLumG, G, fma_(-kLumR, R, L)) * (1.0f / kLumB), 0.0f);
    return make_float4(R, G, B, V);
}

// ------------------------------------------------------------------------- direct (pair arithmetic)
// One thread per pixel, taps from global memory, the arithmetic of the pair kernel with T = float.
__global__ __launch_bounds__(256) void atrous_direct2_kernel(AtrousArgs a)
{
    const Geom g = a.g;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= a.row1) return;
    const int s = a.step;
    const size_t i = pix_index(g, x, y);
    const float4 cc = to_lrgv(a.in[i]);
    const float4 cn = a.nd[i];

    float var_c;
    const bool okl = x - 1 >= 0, okr = x + 1 < g.W, oku = y - 1 >= 0, okd = y + 1 < g.H;
    auto var_at = [&](int tx, int ty) { return a.in[pix_index(g, tx, ty)].w; };
    if (okl && okr && oku && okd) {
        var_c = prefilter9(var_at(x - 1, y - 1), var_at(x - 1, y), var_at(x - 1, y + 1), var_at(x, y - 1), cc.w,
                           var_at(x, y + 1), var_at(x + 1, y - 1), var_at(x + 1, y), var_at(x + 1, y + 1));
    } else {
        float gs = 0.25f, vs = 0.25f * cc.w;
        if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x - 1, y - 1), vs); }
        if (okl)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x - 1, y), vs); }
        if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x - 1, y + 1), vs); }
        if (oku)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x, y - 1), vs); }
        if (okd)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x, y + 1), vs); }
        if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x + 1, y - 1), vs); }
        if (okr)        { gs += 0.125f;  vs = fma_(0.125f, var_at(x + 1, y), vs); }
        if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, var_at(x + 1, y + 1), vs); }
        var_c = vs / gs;
    }
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    const float gz = fabsf(a.nd[pix_index(g, x1, y)].w - cn.w) + fabsf(a.nd[pix_index(g, x, y1)].w - cn.w);
    CenterT<float> k;
    k.nx = cn.x; k.ny = cn.y; k.nz = cn.z; k.z = cn.w; k.l = cc.x;
    k.zc = is_zero3(cn) ? 1.0f : 0.0f;
    center_consts<float>(k, var_c, gz, a.sigma_z, a.sigma_l, (float)s);

    AccT<float> acc = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            if (dx == 0 && dy == 0) { tap_center<float>(acc, cc.x, cc.y, cc.z, cc.w); continue; }
            const int tx = x + s * dx, ty = y + s * dy;
            if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) continue;
            const size_t ti = pix_index(g, tx, ty);
            const float4 tc = to_lrgv(a.in[ti]);
            const float4 tn = a.nd[ti];
            const TapT<float> t = { tc.x, tc.y, tc.z, tc.w, tn.x, tn.y, tn.z, tn.w };
            const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            tap_eval<float, true>(acc, k, t, kLogB3[adx] + kLogB3[ady], len_class(adx, ady), a.sigma_n);
        }
    }
    a.out[i] = finish2(acc.sw, acc.sl, acc.sr, acc.sg, acc.sv, cc.x, cc.y, cc.z, cc.w);
}

// ------------------------------------------------------------------------- pair stream kernel
template <int S>
struct PairCfg {
    static constexpr int CW = 128;                    // output columns per workgroup: 64 lanes x 2 pixels
    static constexpr int PW = CW + 4 * S;             // staged row width (halo 2S each side), even
    static constexpr int NR = 8, ADV = 4;             // ring rows j-2 .. j+5; a step yields rows j .. j+3 (one per wave)
    // A staged row is 4 "pair planes" (l,r) (g,var) (nx,ny) (nz,z); a plane is PW/2 elements of 16 bytes,
    // element m = [a(2m) a(2m+1) b(2m) b(2m+1)]: one conflict-free ds_read_b128 per lane yields the
    // register pairs (a, a') and (b, b') of the two pixels of a lane for two of the eight tap values.
    static constexpr int PLANE = PW * 8;              // bytes of one pair plane of one staged row
    static constexpr int ROW_BYTES = 4 * PLANE;
    static constexpr int RING_BYTES = NR * ROW_BYTES;
    static constexpr int AUXW = CW + 4;               // aux row: columns x0-2 .. x0+CW+1 (pairs stay 8-byte aligned)
    static constexpr int AUX_ROW = AUXW * 4;
    static constexpr int AUX_OFF = RING_BYTES;        // per output row of the step: var(y-1), var(y+1)
    static constexpr int LDS_BYTES = AUX_OFF + ADV * 2 * AUX_ROW;
    static constexpr int WG_PER_CU = 3;
};

// volatile: keeps the compiler from fusing or hoisting the tap reads (the address space is spelled out: a
// volatile access through a generic pointer would become a flat load)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) f4v lds_quad;
typedef const volatile __attribute__((address_space(3))) f2 lds_pair;
typedef const __attribute__((address_space(3))) float lds_scalar;
struct TwoPairs { f2 a, b; };
__device__ __forceinline__ TwoPairs lds_quad_at(const unsigned char* lds, int off)
{
    const f4v q = *(lds_quad*)(lds + off);
    return TwoPairs{ f2{ q.x, q.y }, f2{ q.z, q.w } };
}
__device__ __forceinline__ f2 lds_f2(const unsigned char* lds, int off) { return *(lds_pair*)(lds + off); }
// two floats 12 bytes apart: the values of columns (c, c+1), c odd, of one of the two planes of a pair plane
// (element m-1 slot 1 and element m slot 0); the compiler fuses the two loads into one ds_read2_b32
__device__ __forceinline__ f2 lds_odd_pair(const unsigned char* lds, int off)
{
    lds_scalar* p = (lds_scalar*)(lds + off);
    return f2{ p[0], p[3] };
}
// two adjacent floats at a 4-byte aligned address (aux rows)
__device__ __forceinline__ f2 lds_f2u(const unsigned char* lds, int off)
{
    lds_scalar* p = (lds_scalar*)(lds + off);
    return f2{ p[0], p[1] };
}

// XE: the strip touches the left / right frame border (per-lane column tests).  Rows are handled the
// same way in both forms, with wave-uniform tests only: rows outside the frame (or the buffer) are
// staged as zeros, and the two lattice rows next to the top / bottom border take the generic prefilter.
template <int S, bool XE>
__device__ __forceinline__ void atrous_pair_body(const AtrousArgs& a, unsigned char* lds, const int tid,
                                                 const int x0, const int ybase, const int jlo, const int jhi)
{
    using C = PairCfg<S>;
    const Geom g = a.g;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = output row of the step
    const int xA = x0 + 2 * lane;                                  // pixels xA, xA + 1
    const bool inA = !XE || xA < g.W, inB = !XE || xA + 1 < g.W;
    const float* in_f = reinterpret_cast<const float*>(a.in);
    const float* nd_f = reinterpret_cast<const float*>(a.nd);
    const int blo = max(g.buf_row0, 0), bhi = min(g.buf_row0 + g.buf_rows, g.H);   // rows that exist

    auto slot_of = [&](const int j) { return (j + 4 * C::NR) % C::NR; };        // j >= -2

    // ---- staging: a refill brings ADV lattice rows (jb .. jb+3); wave wv stages row jb + wv, lane ->
    // columns lane, lane + 64, lane + 128 of the PW staged columns.  Everything that depends on the row is
    // wave-uniform (scalar unit), everything that depends on the lane is a loop constant: a refill costs
    // no vector address arithmetic.  Loads are UNCONDITIONAL at clamped addresses (a load inside a branch
    // makes the compiler drain the whole prefetch queue with vmcnt(0) at the join).  A pixel outside the
    // frame or the buffer only has to present a zero normal: its taps then weigh exactly 0 (log2 0 = -inf)
    // and the finite colour / depth values fetched from the clamped address are multiplied by that 0.
    static_assert(C::PW > 128 && C::PW <= 192, "three 64-lane passes cover a staged row");
    int gofs[3];                   // pixel offset inside a row of the frame (clamped to the frame in the XE form)
    float keepx[3];                // XE: 0 for a column outside the frame
    int sofs[3];                   // byte offset of the staged column inside a pair plane (element col/2, slot col&1)
    bool sact[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int col = lane + 64 * q;
        sact[q] = col < C::PW;
        const int colc = min(col, C::PW - 1);
        const int gx = x0 - 2 * S + colc;
        keepx[q] = (!XE || (gx >= 0 && gx < g.W)) ? 1.0f : 0.0f;
        gofs[q] = XE ? min(max(gx, 0), g.W - 1) : gx;
        sofs[q] = (colc >> 1) * 16 + (colc & 1) * 4;
    }
    float4 pc[3], pn[3];
    bool prow_ok = true;           // wave-uniform: the row being prefetched exists
    const float4 *prc = a.in, *prn = a.nd;
    auto begin_rows = [&](const int jb) {          // scalar part of a refill: row pointers
        const int y = ybase + (jb + wv) * S;
        prow_ok = y >= blo && y < bhi;
        const int yc = min(max(y, blo), bhi - 1);
        prc = a.in + (size_t)(yc - g.buf_row0) * (size_t)g.W;
        prn = a.nd + (size_t)(yc - g.buf_row0) * (size_t)g.W;
    };
    auto load_rows_piece = [&](const int i) {      // i = 0..5: one vector-memory instruction each
        if (i & 1) pn[i >> 1] = prn[gofs[i >> 1]]; else pc[i >> 1] = prc[gofs[i >> 1]];
    };
    auto store_rows = [&](const int jb) {
        unsigned char* row = lds + slot_of(jb + wv) * C::ROW_BYTES;
        const float krow = prow_ok ? 1.0f : 0.0f;
        constexpr int PF = C::PLANE / 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (sact[q]) {
                float* p = reinterpret_cast<float*>(row + sofs[q]);
                float nx = pn[q].x, ny = pn[q].y, nz = pn[q].z;
                if (XE || !prow_ok) { const float kk = XE ? krow * keepx[q] : krow; nx *= kk; ny *= kk; nz *= kk; }
                p[0 * PF] = lum3(pc[q].x, pc[q].y, pc[q].z); p[0 * PF + 2] = pc[q].x;
                p[1 * PF] = pc[q].y;                          p[1 * PF + 2] = pc[q].w;
                p[2 * PF] = nx;                               p[2 * PF + 2] = ny;
                p[3 * PF] = nz;                               p[3 * PF + 2] = pn[q].w;
            }
        }
    };
    // aux rows of the outputs of step jo (wave wv serves its own output row): variance of rows y-1, y+1 at
    // columns x0-2 .. x0+CW+1 (lane -> two columns, lanes 0..3 the last four), staged through LDS for the
    // 3x3 prefilter; z of row y+1 at the lane's own two pixels stays in registers (depth gradient).
    // Unconditional loads at clamped rows / columns: the prefilter and the gradient test which of these
    // values exist (oku / okd / okl / okr), rows of dropped outputs read something harmless.
    // lane -> ONE column per gather (16-byte lane stride: a gather instruction then touches 1 KB, not 2)
    int aofs[3], zofs[2];          // float offsets inside a frame row: var of columns x0-2+lane, +64, the 4 tail columns; z of xA, xA+1
    {
        const int g0 = x0 - 2 + lane, g1 = g0 + 64, g2 = x0 + C::CW - 2 + (lane & 3);
        aofs[0] = (XE ? min(max(g0, 0), g.W - 1) : g0) * 4 + 3;
        aofs[1] = (XE ? min(max(g1, 0), g.W - 1) : g1) * 4 + 3;
        aofs[2] = (XE ? min(max(g2, 0), g.W - 1) : g2) * 4 + 3;
        zofs[0] = (XE ? min(xA, g.W - 1) : xA) * 4 + 3;
        zofs[1] = (XE ? min(xA + 1, g.W - 1) : xA + 1) * 4 + 3;
    }
    float pau[2][2], pax[2];
    f2 zd_cur = f2{ 0.0f, 0.0f }, zd_next = zd_cur;
    const float *pup = in_f, *pdn = in_f, *pzr = nd_f;
    auto begin_aux = [&](const int jo) {
        const int y = min(max(ybase + (jo + wv) * S, blo), bhi - 1);
        const int yu = max(y - 1, blo), yd = min(y + 1, bhi - 1);
        pup = in_f + (size_t)(yu - g.buf_row0) * (size_t)g.W * 4;
        pdn = in_f + (size_t)(yd - g.buf_row0) * (size_t)g.W * 4;
        pzr = nd_f + (size_t)(yd - g.buf_row0) * (size_t)g.W * 4;
    };
    auto load_aux_piece = [&](const int i) {       // i = 0..7: one vector-memory instruction each
        switch (i) {
            case 0: pau[0][0] = pup[aofs[0]]; break;
            case 1: pau[1][0] = pup[aofs[1]]; break;
            case 2: pax[0] = pup[aofs[2]]; break;
            case 3: pau[0][1] = pdn[aofs[0]]; break;
            case 4: pau[1][1] = pdn[aofs[1]]; break;
            case 5: pax[1] = pdn[aofs[2]]; break;
            case 6: zd_next.x = pzr[zofs[0]]; break;
            default: zd_next.y = pzr[zofs[1]]; break;
        }
    };
    auto store_aux = [&]() {
        float* ax = reinterpret_cast<float*>(lds + C::AUX_OFF + wv * 2 * C::AUX_ROW);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            ax[k * C::AUXW + lane] = pau[0][k];
            ax[k * C::AUXW + 64 + lane] = pau[1][k];
            if (lane < 4) ax[k * C::AUXW + C::CW + lane] = pax[k];
        }
        zd_cur = zd_next;
    };

    float4 outA, outB;
    // ---- one step: this wave produces lattice row jw = j + wv, this lane its pixels xA, xA+1
    auto compute = [&](const int j) {
        const int jw = j + wv;
        const int y = ybase + jw * S;
        // The 14 vector-memory instructions of the next refill (6 row loads, 8 aux gathers) are issued ONE PER
        // TAP inside the tap loop: issued together at the top of a step they queue up in the CU's one texture
        // address unit (4 waves x 14 at once after the barrier) and every wave stalls at issue -- a quarter of
        // the step.  They are unconditional (clamped rows): the last step fetches rows it does not use.
        begin_rows(j + 6);
        begin_aux(j + C::ADV);
        int rb[5];
#pragma unroll
        for (int tr = 0; tr < 5; ++tr) rb[tr] = slot_of(jw - 2 + tr) * C::ROW_BYTES + (S + lane) * 16;   // element of columns xA, xA+1 of tap row tr
        const int cb = rb[2];
        CenterT<f2> k;
        const f2 cv = lds_f2(lds, cb + 1 * C::PLANE + 8);         // (l, r, g are re-read where they are needed: centre tap, finish)
        { const TwoPairs q = lds_quad_at(lds, cb + 2 * C::PLANE); k.nx = q.a; k.ny = q.b; }
        { const TwoPairs q = lds_quad_at(lds, cb + 3 * C::PLANE); k.nz = q.a; k.z = q.b; }
        k.l = lds_f2(lds, cb + 0 * C::PLANE);
        const bool zA = k.nx.x == 0.0f && k.ny.x == 0.0f && k.nz.x == 0.0f, zB = k.nx.y == 0.0f && k.ny.y == 0.0f && k.nz.y == 0.0f;
        k.zc = f2{ zA ? 1.0f : 0.0f, zB ? 1.0f : 0.0f };

        // A.A.1: 3x3 prefilter of the variance.  Columns xA-1 .. xA+2 of rows y-1 (aux), y (ring), y+1 (aux)
        const unsigned char* axb = lds + C::AUX_OFF + wv * 2 * C::AUX_ROW + lane * 8;    // aux column xA-2
        const f2 vu_l = lds_f2u(axb, 0 * C::AUX_ROW + 4), vu_c = lds_f2(axb, 0 * C::AUX_ROW + 8), vu_r = lds_f2u(axb, 0 * C::AUX_ROW + 12);
        const f2 vd_l = lds_f2u(axb, 1 * C::AUX_ROW + 4), vd_c = lds_f2(axb, 1 * C::AUX_ROW + 8), vd_r = lds_f2u(axb, 1 * C::AUX_ROW + 12);
        const f2 vm_l = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 - 12), vm_r = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 + 4);   // var of (xA-1, xA), (xA+1, xA+2)
        f2 zr = lds_odd_pair(lds, cb + 3 * C::PLANE + 8 + 4);     // z of columns xA+1, xA+2
        const f2 zd = zd_cur;                                     // z of row y+1, columns xA, xA+1
        const bool oku = y - 1 >= 0, okd = y + 1 < g.H;           // wave-uniform
        f2 var_c;
        if (!XE && oku && okd) {
            f2 v = splat(0.0625f, f2{}) * vu_l;
            v = fma_(splat(0.125f, f2{}), vm_l, v);  v = fma_(splat(0.0625f, f2{}), vd_l, v);
            v = fma_(splat(0.125f, f2{}), vu_c, v);  v = fma_(splat(0.25f, f2{}), cv, v);   v = fma_(splat(0.125f, f2{}), vd_c, v);
            v = fma_(splat(0.0625f, f2{}), vu_r, v); v = fma_(splat(0.125f, f2{}), vm_r, v); v = fma_(splat(0.0625f, f2{}), vd_r, v);
            var_c = v;
        } else {
            // frame borders: renormalised prefilter (same operations as the direct kernel)
            float vcs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int x = xA + i;
                const bool okl = !XE || x - 1 >= 0, okr = !XE || x + 1 < g.W;
                const float ul = i ? vu_l.y : vu_l.x, uc = i ? vu_c.y : vu_c.x, ur = i ? vu_r.y : vu_r.x;
                const float dl = i ? vd_l.y : vd_l.x, dc = i ? vd_c.y : vd_c.x, dr = i ? vd_r.y : vd_r.x;
                const float ml = i ? vm_l.y : vm_l.x, mr = i ? vm_r.y : vm_r.x, mc = i ? cv.y : cv.x;
                if (okl && okr && oku && okd) { vcs[i] = prefilter9(ul, ml, dl, uc, mc, dc, ur, mr, dr); continue; }
                float gs = 0.25f, vs = 0.25f * mc;
                if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, ul, vs); }
                if (okl)        { gs += 0.125f;  vs = fma_(0.125f, ml, vs); }
                if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, dl, vs); }
                if (oku)        { gs += 0.125f;  vs = fma_(0.125f, uc, vs); }
                if (okd)        { gs += 0.125f;  vs = fma_(0.125f, dc, vs); }
                if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, ur, vs); }
                if (okr)        { gs += 0.125f;  vs = fma_(0.125f, mr, vs); }
                if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, dr, vs); }
                vcs[i] = vs / gs;
            }
            var_c = f2{ vcs[0], vcs[1] };
        }
        if (XE) {
            if (xA + 1 >= g.W) zr.x = k.z.x;
            if (xA + 2 >= g.W) zr.y = k.z.y;
        }
        const f2 gz = f2{ fabsf(zr.x - k.z.x) + fabsf(zd.x - k.z.x), fabsf(zr.y - k.z.y) + fabsf(zd.y - k.z.y) };
        center_consts<f2>(k, var_c, gz, a.sigma_z, a.sigma_l, (float)S);

        bool rowv[5];
#pragma unroll
        for (int tr = 0; tr < 5; ++tr) {
            const int yy = y + (tr - 2) * S;
            rowv[tr] = yy >= 0 && yy < g.H;
        }
        // the zero-aware body only where a VALID centre of this wave has a zero normal
        const bool any_zero = __builtin_amdgcn_ballot_w64((zA && inA) || (zB && inB)) != 0ull;
        float sn = a.sigma_n;
        asm volatile("v_mov_b32 %0, %1" : "=v"(sn) : "s"(a.sigma_n));   // VGPR on purpose: leaves the one SGPR operand slot of v_pk_fma_f32 to log2 k
        const f2 sn2 = splat(sn, f2{});
        AccT<f2> acc = { f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 } };

        auto load_tap = [&](const int ti, TapT<f2>& t) {
            const int dxi = ti / 5, tr = ti % 5;
            const int sdx = (dxi - 2) * S;                        // column offset of the tap
            if ((sdx & 1) == 0) {
                const int off = rb[tr] + (sdx / 2) * 16;
                { const TwoPairs q = lds_quad_at(lds, off + 0 * C::PLANE); t.l = q.a; t.r = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 1 * C::PLANE); t.g = q.a; t.v = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 2 * C::PLANE); t.nx = q.a; t.ny = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 3 * C::PLANE); t.nz = q.a; t.z = q.b; }
            } else {          // S = 1, dx = -1 / +1: the pair (xA+dx, xA+dx+1) starts on an odd column
                const int off = rb[tr] + ((sdx - 1) / 2) * 16 + 4;    // slot 1 of the element that holds column xA+dx
                t.l = lds_odd_pair(lds, off + 0 * C::PLANE);  t.r = lds_odd_pair(lds, off + 0 * C::PLANE + 8);
                t.g = lds_odd_pair(lds, off + 1 * C::PLANE);  t.v = lds_odd_pair(lds, off + 1 * C::PLANE + 8);
                t.nx = lds_odd_pair(lds, off + 2 * C::PLANE); t.ny = lds_odd_pair(lds, off + 2 * C::PLANE + 8);
                t.nz = lds_odd_pair(lds, off + 3 * C::PLANE); t.z = lds_odd_pair(lds, off + 3 * C::PLANE + 8);
            }
        };
        auto taps = [&](auto zero_aware) {
            constexpr bool ZA = decltype(zero_aware)::value;
            auto weigh = [&](const int ti, const TapT<f2>& t) {
                const int dx = ti / 5 - 2, dy = ti % 5 - 2;
                const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
                if (dx == 0 && dy == 0) {
                    const TwoPairs lr = lds_quad_at(lds, cb + 0 * C::PLANE), gv = lds_quad_at(lds, cb + 1 * C::PLANE);
                    tap_center<f2>(acc, lr.a, lr.b, gv.a, gv.b);
                    return;
                }
                // Out-of-frame taps are staged as zeros: a zero tap normal gives cos = 0, log2 = -inf and
                // weight exactly 0 for a non-zero centre.  A zero centre would accept them (zero pairs with
                // zero), so the zero-aware body masks them explicitly.
                f2 e0 = splat(kLogB3[adx] + kLogB3[ady], f2{});
                if (ZA) {
                    bool okA = rowv[dy + 2], okB = okA;
                    if (XE) {
                        okA = okA && xA + dx * S >= 0 && xA + dx * S < g.W;
                        okB = okB && xA + 1 + dx * S >= 0 && xA + 1 + dx * S < g.W;
                    }
                    e0 = f2{ okA ? e0.x : kNegInf, okB ? e0.y : kNegInf };
                }
                tap_eval<f2, ZA>(acc, k, t, e0, len_class(adx, ady), sn2);
            };
            // the 24 off-centre taps in dx-outer / dy-inner order, double-buffered: the reads of tap q+1 are
            // issued before tap q is weighted; the centre tap (needs no fetch) sits between q = 11 and 12
            TapT<f2> t0, t1;
            load_tap(0, t0);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const int ti = q < 12 ? q : q + 1, tn = q + 1 < 12 ? q + 1 : q + 2;
                if (q == 12) weigh(12, t0);
                // the reads of the next tap go out BEFORE anything of this tap is weighted (left to itself the
                // scheduler sinks them behind the logarithms, half a tap before their use)
                if (q < 23) { if (q & 1) load_tap(tn, t0); else load_tap(tn, t1); }
                if (q < 6) load_rows_piece(q); else if (q < 14) load_aux_piece(q - 6);
                __builtin_amdgcn_sched_barrier(0);
                if (q & 1) weigh(ti, t1); else weigh(ti, t0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (any_zero) taps(std::true_type{}); else taps(std::false_type{});
        const TwoPairs flr = lds_quad_at(lds, cb + 0 * C::PLANE), fgv = lds_quad_at(lds, cb + 1 * C::PLANE);
        const f2 fl = flr.a, fr = flr.b, fg = fgv.a, fv = fgv.b;
        outA = finish2(acc.sw.x, acc.sl.x, acc.sr.x, acc.sg.x, acc.sv.x, fl.x, fr.x, fg.x, fv.x);
        outB = finish2(acc.sw.y, acc.sl.y, acc.sr.y, acc.sg.y, acc.sv.y, fl.y, fr.y, fg.y, fv.y);
    };
    auto write_out = [&](const int j) {
        const int jw = j + wv;
        if (jw >= jlo && jw < jhi) {
            float4* o = a.out + (size_t)(ybase + jw * S - g.buf_row0) * (size_t)g.W + (size_t)xA;
            if (inA) o[0] = outA;
            if (inB) o[1] = outB;
        }
    };

    // ---- prologue: ring rows j0-2 .. j0+5 and the aux rows of the first step
    const int j0 = jlo & ~3;
    auto load_rows = [&](const int jb) { begin_rows(jb); for (int i = 0; i < 6; ++i) load_rows_piece(i); };
    load_rows(j0 - 2);
    store_rows(j0 - 2);
    load_rows(j0 + 2);
    begin_aux(j0);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_aux_piece(i);
    store_rows(j0 + 2);
    store_aux();
    __syncthreads();

    const int span = jhi - j0;
    const int jq1 = j0 + span * RMD_PRIO_T1 / 16, jq2 = j0 + span * RMD_PRIO_T2 / 16, jq3 = j0 + span * RMD_PRIO_T3 / 16;
    __builtin_amdgcn_s_setprio(3);
#ifdef RMD_ATROUS_TRACE
    unsigned long long ph[5] = { 0, 0, 0, 0, 0 }, tp = __builtin_amdgcn_s_memtime();
#endif
    for (int j = j0; j < jhi; j += C::ADV) {
        if (j >= jq3)      __builtin_amdgcn_s_setprio(0);
        else if (j >= jq2) __builtin_amdgcn_s_setprio(1);
        else if (j >= jq1) __builtin_amdgcn_s_setprio(2);
        const bool more = j + C::ADV < jhi;
        RMD_PHASE(0)
        compute(j);
        RMD_PHASE(1)
        if (!more) { write_out(j); break; }
        __syncthreads();                                     // every wave is done reading rows j-2 .. j+1 and the aux rows
        RMD_PHASE(2)
        store_rows(j + 6); store_aux();
        write_out(j);
        RMD_PHASE(3)
        __syncthreads();
        RMD_PHASE(4)
    }
#ifdef RMD_ATROUS_TRACE
    if (tid == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 5; ++i) g_atrous_phase[8 * blockIdx.x + i] = ph[i];
#endif
}

template <int S>
__global__ __launch_bounds__(256, 3) void atrous_pair_kernel(AtrousArgs a)
{
    using C = PairCfg<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char atrous_lds[];
    const int tid = threadIdx.x;
    const int pid = blockIdx.x;
    // XCD-aware order: workgroups pid, pid+8, ... share an XCD (round-robin dispatch).  Each XCD gets one
    // contiguous run of the interior work (neighbouring strips / lattices share halo columns and the +-1
    // variance rows in its L2) followed by its share of the border-strip work.
    const int xcd = pid & (kXcds - 1), slot = pid >> 3;
    int L;
    if (slot < a.int_per_xcd) { L = xcd * a.int_per_xcd + slot; if (L >= a.total_int) return; }
    else { L = a.total_int + xcd * a.xe_per_xcd + (slot - a.int_per_xcd); if (L >= a.nblocks) return; }
    // interior strips first (band height band_h), then the border strips (band_h_xe)
    int strip, band, r, bh;
    if (L < a.total_int) {
        r = L % S; const int t = L / S;
        strip = a.xe_lo + t % a.n_int; band = t / a.n_int; bh = a.band_h;
    } else {
        const int Lx = L - a.total_int, nxe = a.nstrips - a.n_int;
        r = Lx % S; const int t = Lx / S, e = t % nxe;
        strip = e < a.xe_lo ? e : a.n_int + e; band = t / nxe; bh = a.band_h_xe;
    }
    const int x0 = strip * C::CW;
    const int yb = a.band_base + band * bh;
    const int lo = max(yb, a.row0), hi = min(yb + bh, a.row1);
    const int ybase = yb + r;
    const int jlo = lo > ybase ? (lo - ybase + S - 1) / S : 0;
    const int jhi = hi > ybase ? (hi - ybase + S - 1) / S : 0;      // exclusive
    if (jlo >= jhi) return;
    const bool xedge = (x0 - 2 * S < 0) || (x0 + C::CW + 2 * S > a.g.W);
#ifdef RMD_ATROUS_TRACE
    const unsigned long long trace_t0 = wall_clock64(), trace_c0 = __builtin_amdgcn_s_memtime();
#endif
    if (xedge) atrous_pair_body<S, true>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
    else       atrous_pair_body<S, false>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
#ifdef RMD_ATROUS_TRACE
    __syncthreads();
    if (tid == 0 && pid < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_atrous_trace[6 * pid + 0] = trace_t0;
        g_atrous_trace[6 * pid + 1] = wall_clock64();
        g_atrous_trace[6 * pid + 2] = ((unsigned long long)xcc << 32) | hw;
        g_atrous_trace[6 * pid + 3] = ((unsigned long long)L << 8) | (xedge ? 1u : 0u);
        g_atrous_trace[6 * pid + 4] = __builtin_amdgcn_s_memtime() - trace_c0;
        g_atrous_trace[6 * pid + 5] = (unsigned long long)(jhi - jlo);
    }
#endif
}

// ------------------------------------------------------------------------- 2 x 2 pixel block per lane (variant 8)
// The pixel-pair arithmetic with TWO output rows per wave: a lane owns pixels (xA, xA+1) of lattice rows jw and
// jw+1, walks the 5 x 6 tap positions both rows see and weighs each position for the rows that tap it (rows 1..4
// of the window for both, row 0 for the upper and row 5 for the lower output only).  One position is four
// ds_read_b128 as before, so an output costs 480 instead of 800 bytes of LDS reads with every instruction still
// packed (the row-pair kernel has the same LDS ratio but weighs its rows 0 and 5 unpacked).  A step yields 8
// lattice rows from a ring of 12; that is 59-70 KB of LDS, i.e. two workgroups = 2 waves per SIMD with up to
// 256 VGPRs each, which the two independent accumulator sets and the refill registers (12 + 16 loads in flight
// per lane) use.  Same per-output operation order as atrous_direct2_kernel: identical bits.
template <int S>
struct QuadCfg {
    using P = PairCfg<S>;
    static constexpr int ADV = 8, NR = 12;           // ring rows j-2 .. j+9; a step yields rows j .. j+7 (two per wave)
    static constexpr int RING_BYTES = NR * P::ROW_BYTES;
    static constexpr int AUX_OFF = RING_BYTES;       // per output row of the step: var(y-1), var(y+1)
    static constexpr int LDS_BYTES = AUX_OFF + ADV * 2 * P::AUX_ROW;
    static constexpr int WG_PER_CU = 2 * LDS_BYTES <= 160 * 1024 ? 2 : 1;
};

template <int S, bool XE>
__device__ __forceinline__ void atrous_quad_body(const AtrousArgs& a, unsigned char* lds, const int tid,
                                                 const int x0, const int ybase, const int jlo, const int jhi)
{
    using C = PairCfg<S>;
    using Q = QuadCfg<S>;
    const Geom g = a.g;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave -> output rows j + 2 wv, j + 2 wv + 1 of the step
    const int xA = x0 + 2 * lane;
    const bool inA = !XE || xA < g.W, inB = !XE || xA + 1 < g.W;
    const float* in_f = reinterpret_cast<const float*>(a.in);
    const float* nd_f = reinterpret_cast<const float*>(a.nd);
    const int blo = max(g.buf_row0, 0), bhi = min(g.buf_row0 + g.buf_rows, g.H);

    auto slot_of = [&](const int j) { return (j + 4 * Q::NR) % Q::NR; };        // j >= -2

    // ---- staging (as in the pixel-pair kernel; a wave stages the two rows jb + 2 wv, jb + 2 wv + 1 of a refill)
    int gofs[3], sofs[3];
    float keepx[3];
    bool sact[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int col = lane + 64 * q;
        sact[q] = col < C::PW;
        const int colc = min(col, C::PW - 1);
        const int gx = x0 - 2 * S + colc;
        keepx[q] = (!XE || (gx >= 0 && gx < g.W)) ? 1.0f : 0.0f;
        gofs[q] = XE ? min(max(gx, 0), g.W - 1) : gx;
        sofs[q] = (colc >> 1) * 16 + (colc & 1) * 4;
    }
    float4 pc[2][3], pn[2][3];
    bool prow_ok[2] = { true, true };
    const float4 *prc[2] = { a.in, a.in }, *prn[2] = { a.nd, a.nd };
    auto begin_rows = [&](const int jb) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = ybase + (jb + 2 * wv + r) * S;
            prow_ok[r] = y >= blo && y < bhi;
            const int yc = min(max(y, blo), bhi - 1);
            prc[r] = a.in + (size_t)(yc - g.buf_row0) * (size_t)g.W;
            prn[r] = a.nd + (size_t)(yc - g.buf_row0) * (size_t)g.W;
        }
    };
    auto load_rows_piece = [&](const int i) {      // i = 0..11: one vector-memory instruction each
        const int r = i / 6, ii = i % 6;
        if (ii & 1) pn[r][ii >> 1] = prn[r][gofs[ii >> 1]]; else pc[r][ii >> 1] = prc[r][gofs[ii >> 1]];
    };
    auto store_rows = [&](const int jb) {
        constexpr int PF = C::PLANE / 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            unsigned char* row = lds + slot_of(jb + 2 * wv + r) * C::ROW_BYTES;
            const float krow = prow_ok[r] ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (sact[q]) {
                    float* p = reinterpret_cast<float*>(row + sofs[q]);
                    float nx = pn[r][q].x, ny = pn[r][q].y, nz = pn[r][q].z;
                    if (XE || !prow_ok[r]) { const float kk = XE ? krow * keepx[q] : krow; nx *= kk; ny *= kk; nz *= kk; }
                    p[0 * PF] = lum3(pc[r][q].x, pc[r][q].y, pc[r][q].z); p[0 * PF + 2] = pc[r][q].x;
                    p[1 * PF] = pc[r][q].y;                                p[1 * PF + 2] = pc[r][q].w;
                    p[2 * PF] = nx;                                        p[2 * PF + 2] = ny;
                    p[3 * PF] = nz;                                        p[3 * PF + 2] = pn[r][q].w;
                }
            }
        }
    };
    int aofs[3], zofs[2];
    {
        const int g0 = x0 - 2 + lane, g1 = g0 + 64, g2 = x0 + C::CW - 2 + (lane & 3);
        aofs[0] = (XE ? min(max(g0, 0), g.W - 1) : g0) * 4 + 3;
        aofs[1] = (XE ? min(max(g1, 0), g.W - 1) : g1) * 4 + 3;
        aofs[2] = (XE ? min(max(g2, 0), g.W - 1) : g2) * 4 + 3;
        zofs[0] = (XE ? min(xA, g.W - 1) : xA) * 4 + 3;
        zofs[1] = (XE ? min(xA + 1, g.W - 1) : xA + 1) * 4 + 3;
    }
    float pau[2][2][2], pax[2][2];                 // [output row][...]
    f2 zd_cur[2] = { f2{ 0.0f, 0.0f }, f2{ 0.0f, 0.0f } }, zd_next[2] = { zd_cur[0], zd_cur[0] };
    const float *pup[2] = { in_f, in_f }, *pdn[2] = { in_f, in_f }, *pzr[2] = { nd_f, nd_f };
    auto begin_aux = [&](const int jo) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = min(max(ybase + (jo + 2 * wv + r) * S, blo), bhi - 1);
            const int yu = max(y - 1, blo), yd = min(y + 1, bhi - 1);
            pup[r] = in_f + (size_t)(yu - g.buf_row0) * (size_t)g.W * 4;
            pdn[r] = in_f + (size_t)(yd - g.buf_row0) * (size_t)g.W * 4;
            pzr[r] = nd_f + (size_t)(yd - g.buf_row0) * (size_t)g.W * 4;
        }
    };
    auto load_aux_piece = [&](const int i) {       // i = 0..15: one vector-memory instruction each
        const int r = i / 8;
        switch (i % 8) {
            case 0: pau[r][0][0] = pup[r][aofs[0]]; break;
            case 1: pau[r][1][0] = pup[r][aofs[1]]; break;
            case 2: pax[r][0] = pup[r][aofs[2]]; break;
            case 3: pau[r][0][1] = pdn[r][aofs[0]]; break;
            case 4: pau[r][1][1] = pdn[r][aofs[1]]; break;
            case 5: pax[r][1] = pdn[r][aofs[2]]; break;
            case 6: zd_next[r].x = pzr[r][zofs[0]]; break;
            default: zd_next[r].y = pzr[r][zofs[1]]; break;
        }
    };
    auto store_aux = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float* ax = reinterpret_cast<float*>(lds + Q::AUX_OFF + (2 * wv + r) * 2 * C::AUX_ROW);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                ax[k * C::AUXW + lane] = pau[r][0][k];
                ax[k * C::AUXW + 64 + lane] = pau[r][1][k];
                if (lane < 4) ax[k * C::AUXW + C::CW + lane] = pax[r][k];
            }
            zd_cur[r] = zd_next[r];
        }
    };

    float4 out[2][2];                              // [output row][pixel]
    auto compute = [&](const int j) {
        const int jw = j + 2 * wv;                 // upper output row; the lower one is jw + 1
        const int y0 = ybase + jw * S;
        begin_rows(j + 10);
        begin_aux(j + Q::ADV);
        int rb[6];                                 // window rows jw-2 .. jw+3
#pragma unroll
        for (int tr = 0; tr < 6; ++tr) rb[tr] = slot_of(jw - 2 + tr) * C::ROW_BYTES + (S + lane) * 16;
        CenterT<f2> k[2];
        bool zero[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int cb = rb[2 + r];
            const int y = y0 + r * S;
            const f2 cv = lds_f2(lds, cb + 1 * C::PLANE + 8);
            { const TwoPairs q = lds_quad_at(lds, cb + 2 * C::PLANE); k[r].nx = q.a; k[r].ny = q.b; }
            { const TwoPairs q = lds_quad_at(lds, cb + 3 * C::PLANE); k[r].nz = q.a; k[r].z = q.b; }
            k[r].l = lds_f2(lds, cb + 0 * C::PLANE);
            const bool zA = k[r].nx.x == 0.0f && k[r].ny.x == 0.0f && k[r].nz.x == 0.0f;
            const bool zB = k[r].nx.y == 0.0f && k[r].ny.y == 0.0f && k[r].nz.y == 0.0f;
            k[r].zc = f2{ zA ? 1.0f : 0.0f, zB ? 1.0f : 0.0f };
            zero[r] = (zA && inA) || (zB && inB);
            const unsigned char* axb = lds + Q::AUX_OFF + (2 * wv + r) * 2 * C::AUX_ROW + lane * 8;
            const f2 vu_l = lds_f2u(axb, 0 * C::AUX_ROW + 4), vu_c = lds_f2(axb, 0 * C::AUX_ROW + 8), vu_r = lds_f2u(axb, 0 * C::AUX_ROW + 12);
            const f2 vd_l = lds_f2u(axb, 1 * C::AUX_ROW + 4), vd_c = lds_f2(axb, 1 * C::AUX_ROW + 8), vd_r = lds_f2u(axb, 1 * C::AUX_ROW + 12);
            const f2 vm_l = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 - 12), vm_r = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 + 4);
            f2 zr = lds_odd_pair(lds, cb + 3 * C::PLANE + 8 + 4);
            const f2 zd = zd_cur[r];
            const bool oku = y - 1 >= 0, okd = y + 1 < g.H;
            f2 var_c;
            if (!XE && oku && okd) {
                f2 v = splat(0.0625f, f2{}) * vu_l;
                v = fma_(splat(0.125f, f2{}), vm_l, v);  v = fma_(splat(0.0625f, f2{}), vd_l, v);
                v = fma_(splat(0.125f, f2{}), vu_c, v);  v = fma_(splat(0.25f, f2{}), cv, v);   v = fma_(splat(0.125f, f2{}), vd_c, v);
                v = fma_(splat(0.0625f, f2{}), vu_r, v); v = fma_(splat(0.125f, f2{}), vm_r, v); v = fma_(splat(0.0625f, f2{}), vd_r, v);
                var_c = v;
            } else {
                float vcs[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int x = xA + i;
                    const bool okl = !XE || x - 1 >= 0, okr = !XE || x + 1 < g.W;
                    const float ul = i ? vu_l.y : vu_l.x, uc = i ? vu_c.y : vu_c.x, ur = i ? vu_r.y : vu_r.x;
                    const float dl = i ? vd_l.y : vd_l.x, dc = i ? vd_c.y : vd_c.x, dr = i ? vd_r.y : vd_r.x;
                    const float ml = i ? vm_l.y : vm_l.x, mr = i ? vm_r.y : vm_r.x, mc = i ? cv.y : cv.x;
                    if (okl && okr && oku && okd) { vcs[i] = prefilter9(ul, ml, dl, uc, mc, dc, ur, mr, dr); continue; }
                    float gs = 0.25f, vs = 0.25f * mc;
                    if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, ul, vs); }
                    if (okl)        { gs += 0.125f;  vs = fma_(0.125f, ml, vs); }
                    if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, dl, vs); }
                    if (oku)        { gs += 0.125f;  vs = fma_(0.125f, uc, vs); }
                    if (okd)        { gs += 0.125f;  vs = fma_(0.125f, dc, vs); }
                    if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, ur, vs); }
                    if (okr)        { gs += 0.125f;  vs = fma_(0.125f, mr, vs); }
                    if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, dr, vs); }
                    vcs[i] = vs / gs;
                }
                var_c = f2{ vcs[0], vcs[1] };
            }
            if (XE) {
                if (xA + 1 >= g.W) zr.x = k[r].z.x;
                if (xA + 2 >= g.W) zr.y = k[r].z.y;
            }
            const f2 gz = f2{ fabsf(zr.x - k[r].z.x) + fabsf(zd.x - k[r].z.x), fabsf(zr.y - k[r].z.y) + fabsf(zd.y - k[r].z.y) };
            center_consts<f2>(k[r], var_c, gz, a.sigma_z, a.sigma_l, (float)S);
        }
        bool rowv[6];
#pragma unroll
        for (int tr = 0; tr < 6; ++tr) {
            const int yy = y0 + (tr - 2) * S;
            rowv[tr] = yy >= 0 && yy < g.H;
        }
        const bool any_zero = __builtin_amdgcn_ballot_w64(zero[0] || zero[1]) != 0ull;
        float sn = a.sigma_n;
        asm volatile("v_mov_b32 %0, %1" : "=v"(sn) : "s"(a.sigma_n));
        const f2 sn2 = splat(sn, f2{});
        AccT<f2> acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = AccT<f2>{ f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 } };

        auto load_pos = [&](const int pi, TapT<f2>& t) {          // position = (column offset, window row)
            const int dxi = pi / 6, tr = pi % 6;
            const int sdx = (dxi - 2) * S;
            if ((sdx & 1) == 0) {
                const int off = rb[tr] + (sdx / 2) * 16;
                { const TwoPairs q = lds_quad_at(lds, off + 0 * C::PLANE); t.l = q.a; t.r = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 1 * C::PLANE); t.g = q.a; t.v = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 2 * C::PLANE); t.nx = q.a; t.ny = q.b; }
                { const TwoPairs q = lds_quad_at(lds, off + 3 * C::PLANE); t.nz = q.a; t.z = q.b; }
            } else {
                const int off = rb[tr] + ((sdx - 1) / 2) * 16 + 4;
                t.l = lds_odd_pair(lds, off + 0 * C::PLANE);  t.r = lds_odd_pair(lds, off + 0 * C::PLANE + 8);
                t.g = lds_odd_pair(lds, off + 1 * C::PLANE);  t.v = lds_odd_pair(lds, off + 1 * C::PLANE + 8);
                t.nx = lds_odd_pair(lds, off + 2 * C::PLANE); t.ny = lds_odd_pair(lds, off + 2 * C::PLANE + 8);
                t.nz = lds_odd_pair(lds, off + 3 * C::PLANE); t.z = lds_odd_pair(lds, off + 3 * C::PLANE + 8);
            }
        };
        auto taps = [&](auto zero_aware) {
            constexpr bool ZA = decltype(zero_aware)::value;
            auto weigh = [&](const int pi, const TapT<f2>& t) {
                const int dx = pi / 6 - 2, tr = pi % 6;
                const int adx = dx < 0 ? -dx : dx;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int dy = tr - 2 - r;
                    if (dy < -2 || dy > 2) continue;
                    const int ady = dy < 0 ? -dy : dy;
                    if (dx == 0 && dy == 0) { tap_center<f2>(acc[r], t.l, t.r, t.g, t.v); continue; }
                    f2 e0 = splat(kLogB3[adx] + kLogB3[ady], f2{});
                    if (ZA) {
                        bool okA = rowv[tr], okB = okA;
                        if (XE) {
                            okA = okA && xA + dx * S >= 0 && xA + dx * S < g.W;
                            okB = okB && xA + 1 + dx * S >= 0 && xA + 1 + dx * S < g.W;
                        }
                        e0 = f2{ okA ? e0.x : kNegInf, okB ? e0.y : kNegInf };
                    }
                    tap_eval<f2, ZA>(acc[r], k[r], t, e0, len_class(adx, ady), sn2);
                }
            };
            // 30 positions, dx outer / window row inner (each output sees its 25 taps in dx-outer / dy-inner order),
            // double-buffered; the 28 vector-memory instructions of the next refill go out one per position
            TapT<f2> t0, t1;
            load_pos(0, t0);
#pragma unroll
            for (int q = 0; q < 30; ++q) {
                if (q < 29) { if (q & 1) load_pos(q + 1, t0); else load_pos(q + 1, t1); }
                if (q < 12) load_rows_piece(q); else if (q < 28) load_aux_piece(q - 12);
                __builtin_amdgcn_sched_barrier(0);
                if (q & 1) weigh(q, t1); else weigh(q, t0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (any_zero) taps(std::true_type{}); else taps(std::false_type{});
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int cb = rb[2 + r];
            const TwoPairs flr = lds_quad_at(lds, cb + 0 * C::PLANE), fgv = lds_quad_at(lds, cb + 1 * C::PLANE);
            out[r][0] = finish2(acc[r].sw.x, acc[r].sl.x, acc[r].sr.x, acc[r].sg.x, acc[r].sv.x, flr.a.x, flr.b.x, fgv.a.x, fgv.b.x);
            out[r][1] = finish2(acc[r].sw.y, acc[r].sl.y, acc[r].sr.y, acc[r].sg.y, acc[r].sv.y, flr.a.y, flr.b.y, fgv.a.y, fgv.b.y);
        }
    };
    auto write_out = [&](const int j) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int jw = j + 2 * wv + r;
            if (jw >= jlo && jw < jhi) {
                float4* o = a.out + (size_t)(ybase + jw * S - g.buf_row0) * (size_t)g.W + (size_t)xA;
                if (inA) o[0] = out[r][0];
                if (inB) o[1] = out[r][1];
            }
        }
    };

    // ---- prologue: ring rows j0-2 .. j0+9 (one and a half refills) and the aux rows of the first step
    const int j0 = jlo & ~7;
    begin_rows(j0 - 2);
#pragma unroll
    for (int i = 0; i < 12; ++i) load_rows_piece(i);
    store_rows(j0 - 2);
    begin_rows(j0 + 6);                            // rows j0+6 .. j0+13: only j0+6 .. j0+9 (waves 0, 1) belong to the first window
#pragma unroll
    for (int i = 0; i < 12; ++i) load_rows_piece(i);
    begin_aux(j0);
#pragma unroll
    for (int i = 0; i < 16; ++i) load_aux_piece(i);
    if (wv < 2) store_rows(j0 + 6);
    store_aux();
    __syncthreads();

    for (int j = j0; j < jhi; j += Q::ADV) {
        const bool more = j + Q::ADV < jhi;
        compute(j);
        if (!more) { write_out(j); break; }
        __syncthreads();                           // every wave is done reading rows j-2 .. j+5 and the aux rows
        store_rows(j + 10); store_aux();
        write_out(j);
        __syncthreads();
    }
}

template <int S>
__global__ __launch_bounds__(256, 2) void atrous_quad_kernel(AtrousArgs a)
{
    using C = PairCfg<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char atrous_lds[];
    const int tid = threadIdx.x;
    const int pid = blockIdx.x;
    const int xcd = pid & (kXcds - 1), slot = pid >> 3;
    int L;
    if (slot < a.int_per_xcd) { L = xcd * a.int_per_xcd + slot; if (L >= a.total_int) return; }
    else { L = a.total_int + xcd * a.xe_per_xcd + (slot - a.int_per_xcd); if (L >= a.nblocks) return; }
    int strip, band, r, bh;
    if (L < a.total_int) {
        r = L % S; const int t = L / S;
        strip = a.xe_lo + t % a.n_int; band = t / a.n_int; bh = a.band_h;
    } else {
        const int Lx = L - a.total_int, nxe = a.nstrips - a.n_int;
        r = Lx % S; const int t = Lx / S, e = t % nxe;
        strip = e < a.xe_lo ? e : a.n_int + e; band = t / nxe; bh = a.band_h_xe;
    }
    const int x0 = strip * C::CW;
    const int yb = a.band_base + band * bh;
    const int lo = max(yb, a.row0), hi = min(yb + bh, a.row1);
    const int ybase = yb + r;
    const int jlo = lo > ybase ? (lo - ybase + S - 1) / S : 0;
    const int jhi = hi > ybase ? (hi - ybase + S - 1) / S : 0;
    if (jlo >= jhi) return;
    const bool xedge = (x0 - 2 * S < 0) || (x0 + C::CW + 2 * S > a.g.W);
    if (xedge) atrous_quad_body<S, true>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
    else       atrous_quad_body<S, false>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
}

// ------------------------------------------------------------------------- loader / consumer form
// The pixel-pair arithmetic with staging taken out of the compute waves (variant 7).  The ablations of the two
// kernels above say a step is a chain -- refill loads, 25 x [LDS reads, arithmetic], barrier, LDS stores,
// barrier -- that overlaps only through the other workgroups of the CU, with the refill one step deep.  Here a
// workgroup is 4 compute waves + 1 LOADER wave:
//   loader    streams "packages" k = 0, 1, 2 ... (lattice row j0-2+k of the strip + the variance rows above and
//             below output row k-4) through a register pipeline three packages deep into the LDS ring, and
//             publishes a monotonic count of finished packages;
//   consumers wait for the count to cover the five rows of their output row, compute it (same code path as the
//             barrier form), store it, and publish how many steps they have finished -- which is what the loader
//             reads before it overwrites a ring slot.  No workgroup barrier after the prologue.
// Every spin is bounded: a protocol bug ends in wrong pixels and an error flag, never in a hang.
#ifndef RMD_LC_DEPTH
#define RMD_LC_DEPTH 3
#endif
template <int S>
struct LcCfg {
    using P = PairCfg<S>;
    // Wave placement decides the shape.  With 128 VGPRs a SIMD holds 4 waves and a workgroup's waves are dealt
    // round-robin over the SIMDs:
    //   4 + 1 waves  three workgroups per CU by LDS, but only two became resident (a SIMD would have needed 6 waves)
    //   6 + 2 waves  two workgroups fill the CU; consumers sit 4 / 4 / 2 / 2 on the SIMDs          <- default
    //   12 + 4 waves one workgroup per CU, 3 consumers + 1 loader on every SIMD (-DRMD_LC_NC=12 -DRMD_LC_NL=4)
    // Measured at 4K, steps 2..16: 6+2 150 us, 12+4 170 us per launch -- and 130 us for the 12+4
    // consumers ALONE (loaders idle, no waits): the arithmetic, not the staging, is what the launch time is made of.
#ifndef RMD_LC_NC
#define RMD_LC_NC 6
#define RMD_LC_NL 2
#endif
    static constexpr int NC = RMD_LC_NC, NL = RMD_LC_NL;   // consumer waves (one output row each per step), loader waves
    static constexpr int THREADS = 64 * (NC + NL);
    static constexpr int WG_PER_CU = 16 / (NC + NL);
    static constexpr int NR = NC == 12 ? 22 : 12;           // ring rows: the NC + 4 of a step + spare
    static constexpr int RING_BYTES = NR * P::ROW_BYTES;
    static constexpr int AUX_OFF = RING_BYTES;       // var(y-1), var(y+1) per consumer wave (a slot belongs to its wave alone)
    static constexpr int FLAG_OFF = AUX_OFF + NC * 2 * P::AUX_ROW;
    static constexpr int LDS_BYTES = FLAG_OFF + 128; // [0..NL) packages committed by loader l, [4..4+NC) steps done by consumer w, [16] error
    static_assert(NL <= 4 && NC <= 12, "flag layout");
    static_assert(WG_PER_CU * LDS_BYTES <= 160 * 1024, "LDS of a CU");
};
__device__ unsigned int g_atrous_lc_errors;          // protocol time-outs (0 in a correct build): rmd_debug_atrous_protocol_errors

typedef volatile __attribute__((address_space(3))) int lds_flag;

template <int S, bool XE>
__device__ __forceinline__ void atrous_lc_body(const AtrousArgs& a, unsigned char* lds, const int tid,
                                               const int x0, const int ybase, const int jlo, const int jhi)
{
    using C = PairCfg<S>;
    using L = LcCfg<S>;
    const Geom g = a.g;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..NC-1 consumers, then the loaders
    const int blo = max(g.buf_row0, 0), bhi = min(g.buf_row0 + g.buf_rows, g.H);
    const int j0 = jlo;
    const int nsteps = (jhi - j0 + L::NC - 1) / L::NC;
    const int nrows = L::NC * nsteps;
    const int npk = nrows + 4;                                     // packages: lattice rows j0-2 .. j0+nrows+1
    lds_flag* flags = (lds_flag*)(lds + L::FLAG_OFF);
    constexpr int kSpinLimit = 1 << 22;
    if (tid < 32) flags[tid] = 0;
    __syncthreads();

    if (wv >= L::NC) {
        // =============================================================== loaders: packages k = ld, ld + NL, ...
        const int ld = wv - L::NC;
#ifndef RMD_LC_PRIO0
        __builtin_amdgcn_s_setprio(3);
#endif
        const float* in_f = reinterpret_cast<const float*>(a.in);
        int gofs[3], sofs[3];
        float keepx[3];
        bool sact[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = lane + 64 * q;
            sact[q] = col < C::PW;
            const int colc = min(col, C::PW - 1);
            const int gx = x0 - 2 * S + colc;
            keepx[q] = (!XE || (gx >= 0 && gx < g.W)) ? 1.0f : 0.0f;
            gofs[q] = XE ? min(max(gx, 0), g.W - 1) : gx;
            sofs[q] = (colc >> 1) * 16 + (colc & 1) * 4;
        }
        struct Pkg { float4 c[3], n[3]; };
        auto issue = [&](const int k, Pkg& p) {                    // 6 vector-memory instructions, unconditional
            const int y = ybase + (j0 - 2 + k) * S;
            const int yc = min(max(y, blo), bhi - 1);
            const float4* rc = a.in + (size_t)(yc - g.buf_row0) * (size_t)g.W;
            const float4* rn = a.nd + (size_t)(yc - g.buf_row0) * (size_t)g.W;
#pragma unroll
#ifdef RMD_LC_ABL_NOLOAD
            for (int q = 0; q < 3; ++q) { p.c[q] = float4{ (float)k, 1.0f, 2.0f, 0.1f }; p.n[q] = float4{ 0.0f, 0.0f, 1.0f, (float)gofs[q] }; }
#else
            for (int q = 0; q < 3; ++q) { p.c[q] = rc[gofs[q]]; p.n[q] = rn[gofs[q]]; }
#endif
        };
        int seen[L::NC] = {};                                      // steps finished by consumer wave w, as last read
#ifdef RMD_ATROUS_TRACE
        unsigned long long ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tp = __builtin_amdgcn_s_memtime();
#endif
        auto commit = [&](const int k, const Pkg& p) {
            RMD_PHASE(7)
            // The ring slot held package k-NR, whose last reader is output row k-NR.  Output row q is step q/NC of
            // wave q%NC and a wave finishes its rows in order: wave w must have finished its last row <= k-NR.
            const int m = k - L::NR;
            if (m >= 0) {
                int need[L::NC];
#pragma unroll
                for (int w = 0; w < L::NC; ++w) {
                    const int qw = m - (m - w + L::NC) % L::NC;
                    need[w] = qw >= 0 ? qw / L::NC + 1 : 0;
                }
                auto behind = [&]() {
                    bool b = false;
#pragma unroll
                    for (int w = 0; w < L::NC; ++w) b = b || seen[w] < need[w];
                    return b;
                };
                int spins = 0;
#ifdef RMD_LC_ABL_NOSYNC
                if (false)
#endif
                while (behind()) {
#pragma unroll
                    for (int w = 0; w < L::NC; ++w) seen[w] = __builtin_amdgcn_readfirstlane(flags[4 + w]);
                    if (!behind()) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > kSpinLimit) { if (lane == 0) { flags[16] = 1; atomicAdd(&g_atrous_lc_errors, 1u); } break; }
                }
            }
            RMD_PHASE(5)
            const int y = ybase + (j0 - 2 + k) * S;
            const float krow = (y >= blo && y < bhi) ? 1.0f : 0.0f;
            unsigned char* row = lds + (k % L::NR) * C::ROW_BYTES;
            constexpr int PF = C::PLANE / 4;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (sact[q]) {
                    float* w = reinterpret_cast<float*>(row + sofs[q]);
                    const float kk = XE ? krow * keepx[q] : krow;
                    w[0 * PF] = lum3(p.c[q].x, p.c[q].y, p.c[q].z); w[0 * PF + 2] = p.c[q].x;
                    w[1 * PF] = p.c[q].y;                            w[1 * PF + 2] = p.c[q].w;
                    w[2 * PF] = p.n[q].x * kk;                       w[2 * PF + 2] = p.n[q].y * kk;
                    w[3 * PF] = p.n[q].z * kk;                       w[3 * PF + 2] = p.n[q].w;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the rows are in LDS before the count says so
            if (lane == 0) flags[ld] = k / L::NL + 1;              // packages this loader has committed
            RMD_PHASE(6)
        };
        constexpr int D = RMD_LC_DEPTH;                            // packages in flight per loader (its registers are the buffer)
        Pkg p[D];
#pragma unroll
        for (int i = 0; i < D; ++i) issue(min(ld + L::NL * i, npk - 1), p[i]);
        for (int k = ld; k < npk; k += L::NL * D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int ki = k + L::NL * i;
                if (ki < npk) commit(ki, p[i]);
                issue(min(ki + L::NL * D, npk - 1), p[i]);
            }
        }
#ifdef RMD_ATROUS_TRACE
        if (lane == 0 && blockIdx.x < 8192 && ld == 0)
            for (int i = 5; i < 8; ++i) g_atrous_phase[8 * blockIdx.x + i] = ph[i];
#endif
        return;
    }

    // ================================================================= consumers
    const int xA = x0 + 2 * lane;
    const bool inA = !XE || xA < g.W, inB = !XE || xA + 1 < g.W;
    const float* nd_f = reinterpret_cast<const float*>(a.nd);
    int zofs[2];
    zofs[0] = (XE ? min(xA, g.W - 1) : xA) * 4 + 3;
    zofs[1] = (XE ? min(xA + 1, g.W - 1) : xA + 1) * 4 + 3;
    auto load_zd = [&](const int q) {                              // z of row y+1 at the lane's own pixels, output row q
        const int y = min(max(ybase + (j0 + q) * S, blo), bhi - 1);
        const float* zr = nd_f + (size_t)(min(y + 1, bhi - 1) - g.buf_row0) * (size_t)g.W * 4;
        return f2{ zr[zofs[0]], zr[zofs[1]] };
    };
    f2 zd_next = load_zd(wv);
    // variance rows above and below the wave's own output row: slot `wv` of the aux ring belongs to this wave alone
    const float* in_f = reinterpret_cast<const float*>(a.in);
    int aofs[3];
    {
        const int g0 = x0 - 2 + lane, g1 = g0 + 64, g2 = x0 + C::CW - 2 + (lane & 3);
        aofs[0] = (XE ? min(max(g0, 0), g.W - 1) : g0) * 4 + 3;
        aofs[1] = (XE ? min(max(g1, 0), g.W - 1) : g1) * 4 + 3;
        aofs[2] = (XE ? min(max(g2, 0), g.W - 1) : g2) * 4 + 3;
    }
    float au[6];
    auto load_aux = [&](const int q) {
        const int yo = min(max(ybase + (j0 + q) * S, blo), bhi - 1);
        const float* up = in_f + (size_t)(max(yo - 1, blo) - g.buf_row0) * (size_t)g.W * 4;
        const float* dn = in_f + (size_t)(min(yo + 1, bhi - 1) - g.buf_row0) * (size_t)g.W * 4;
        au[0] = up[aofs[0]]; au[1] = up[aofs[1]]; au[2] = up[aofs[2]];
        au[3] = dn[aofs[0]]; au[4] = dn[aofs[1]]; au[5] = dn[aofs[2]];
    };
    auto store_aux = [&]() {
        float* ax = reinterpret_cast<float*>(lds + L::AUX_OFF + wv * 2 * C::AUX_ROW);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            ax[r * C::AUXW + lane] = au[3 * r + 0];
            ax[r * C::AUXW + 64 + lane] = au[3 * r + 1];
            if (lane < 4) ax[r * C::AUXW + C::CW + lane] = au[3 * r + 2];
        }
    };
    load_aux(wv);
    store_aux();
    int seen_ready[L::NL] = {};
#ifdef RMD_ATROUS_TRACE
    unsigned long long ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tp = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < nsteps; ++t) {
        const int q = L::NC * t + wv;                              // output row (package numbering: lattice row j0 + q)
        const int jw = j0 + q;
        const int y = ybase + jw * S;
        const f2 zd = zd_next;
#ifndef RMD_LC_ABL_NOOUT
        zd_next = load_zd(min(q + L::NC, nrows - 1));
        load_aux(min(q + L::NC, nrows - 1));
#endif
        RMD_PHASE(2)
        {
            // main rows q .. q+4 = packages 0 .. q+4: (q + 4 - l) / NL + 1 of them come from loader l
            int need[L::NL];
#pragma unroll
            for (int l = 0; l < L::NL; ++l) need[l] = (q + 4 - l + L::NL) / L::NL;
            auto behind = [&]() {
                bool b = false;
#pragma unroll
                for (int l = 0; l < L::NL; ++l) b = b || seen_ready[l] < need[l];
                return b;
            };
            int spins = 0;
#ifdef RMD_LC_ABL_NOSYNC
            if (false)
#endif
            while (behind()) {
#pragma unroll
                for (int l = 0; l < L::NL; ++l) seen_ready[l] = __builtin_amdgcn_readfirstlane(flags[l]);
                if (!behind()) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { if (lane == 0) { flags[16] = 1; atomicAdd(&g_atrous_lc_errors, 1u); } break; }
            }
        }
        RMD_PHASE(0)
        int rb[5];
#pragma unroll
        for (int tr = 0; tr < 5; ++tr) rb[tr] = ((q + tr) % L::NR) * C::ROW_BYTES + (S + lane) * 16;
        const int cb = rb[2];
        CenterT<f2> k;
        const f2 cv = lds_f2(lds, cb + 1 * C::PLANE + 8);
        { const TwoPairs u = lds_quad_at(lds, cb + 2 * C::PLANE); k.nx = u.a; k.ny = u.b; }
        { const TwoPairs u = lds_quad_at(lds, cb + 3 * C::PLANE); k.nz = u.a; k.z = u.b; }
        k.l = lds_f2(lds, cb + 0 * C::PLANE);
        const bool zA = k.nx.x == 0.0f && k.ny.x == 0.0f && k.nz.x == 0.0f, zB = k.nx.y == 0.0f && k.ny.y == 0.0f && k.nz.y == 0.0f;
        k.zc = f2{ zA ? 1.0f : 0.0f, zB ? 1.0f : 0.0f };
        const unsigned char* axb = lds + L::AUX_OFF + wv * 2 * C::AUX_ROW + lane * 8;
        const f2 vu_l = lds_f2u(axb, 0 * C::AUX_ROW + 4), vu_c = lds_f2(axb, 0 * C::AUX_ROW + 8), vu_r = lds_f2u(axb, 0 * C::AUX_ROW + 12);
        const f2 vd_l = lds_f2u(axb, 1 * C::AUX_ROW + 4), vd_c = lds_f2(axb, 1 * C::AUX_ROW + 8), vd_r = lds_f2u(axb, 1 * C::AUX_ROW + 12);
        const f2 vm_l = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 - 12), vm_r = lds_odd_pair(lds, cb + 1 * C::PLANE + 8 + 4);
        f2 zr = lds_odd_pair(lds, cb + 3 * C::PLANE + 8 + 4);
        const bool oku = y - 1 >= 0, okd = y + 1 < g.H;
        // (the six aux reads above have returned before their values are used below; the slot is rewritten at the
        //  end of the step, in program order of this wave)
        f2 var_c;
        if (!XE && oku && okd) {
            f2 v = splat(0.0625f, f2{}) * vu_l;
            v = fma_(splat(0.125f, f2{}), vm_l, v);  v = fma_(splat(0.0625f, f2{}), vd_l, v);
            v = fma_(splat(0.125f, f2{}), vu_c, v);  v = fma_(splat(0.25f, f2{}), cv, v);   v = fma_(splat(0.125f, f2{}), vd_c, v);
            v = fma_(splat(0.0625f, f2{}), vu_r, v); v = fma_(splat(0.125f, f2{}), vm_r, v); v = fma_(splat(0.0625f, f2{}), vd_r, v);
            var_c = v;
        } else {
            float vcs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int x = xA + i;
                const bool okl = !XE || x - 1 >= 0, okr = !XE || x + 1 < g.W;
                const float ul = i ? vu_l.y : vu_l.x, uc = i ? vu_c.y : vu_c.x, ur = i ? vu_r.y : vu_r.x;
                const float dl = i ? vd_l.y : vd_l.x, dc = i ? vd_c.y : vd_c.x, dr = i ? vd_r.y : vd_r.x;
                const float ml = i ? vm_l.y : vm_l.x, mr = i ? vm_r.y : vm_r.x, mc = i ? cv.y : cv.x;
                if (okl && okr && oku && okd) { vcs[i] = prefilter9(ul, ml, dl, uc, mc, dc, ur, mr, dr); continue; }
                float gs = 0.25f, vs = 0.25f * mc;
                if (okl && oku) { gs += 0.0625f; vs = fma_(0.0625f, ul, vs); }
                if (okl)        { gs += 0.125f;  vs = fma_(0.125f, ml, vs); }
                if (okl && okd) { gs += 0.0625f; vs = fma_(0.0625f, dl, vs); }
                if (oku)        { gs += 0.125f;  vs = fma_(0.125f, uc, vs); }
                if (okd)        { gs += 0.125f;  vs = fma_(0.125f, dc, vs); }
                if (okr && oku) { gs += 0.0625f; vs = fma_(0.0625f, ur, vs); }
                if (okr)        { gs += 0.125f;  vs = fma_(0.125f, mr, vs); }
                if (okr && okd) { gs += 0.0625f; vs = fma_(0.0625f, dr, vs); }
                vcs[i] = vs / gs;
            }
            var_c = f2{ vcs[0], vcs[1] };
        }
        if (XE) {
            if (xA + 1 >= g.W) zr.x = k.z.x;
            if (xA + 2 >= g.W) zr.y = k.z.y;
        }
        const f2 gz = f2{ fabsf(zr.x - k.z.x) + fabsf(zd.x - k.z.x), fabsf(zr.y - k.z.y) + fabsf(zd.y - k.z.y) };
        center_consts<f2>(k, var_c, gz, a.sigma_z, a.sigma_l, (float)S);
        bool rowv[5];
#pragma unroll
        for (int tr = 0; tr < 5; ++tr) {
            const int yy = y + (tr - 2) * S;
            rowv[tr] = yy >= 0 && yy < g.H;
        }
        const bool any_zero = __builtin_amdgcn_ballot_w64((zA && inA) || (zB && inB)) != 0ull;
        float sn = a.sigma_n;
        asm volatile("v_mov_b32 %0, %1" : "=v"(sn) : "s"(a.sigma_n));
        const f2 sn2 = splat(sn, f2{});
        AccT<f2> acc = { f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 }, f2{ 0, 0 } };
        auto load_tap = [&](const int ti, TapT<f2>& tp) {
#ifdef RMD_LC_ABL_NOLDS
            // ablation: no LDS reads -- the tap values are register values the compiler cannot see through
            tp.l = k.l; tp.r = k.nx; tp.g = k.ny; tp.v = k.nz; tp.nx = k.nx; tp.ny = k.ny; tp.nz = k.nz; tp.z = k.z;
            asm volatile("" : "+v"(tp.l), "+v"(tp.r), "+v"(tp.g), "+v"(tp.v), "+v"(tp.nx), "+v"(tp.ny), "+v"(tp.nz), "+v"(tp.z));
            return;
#endif
            const int dxi = ti / 5, tr = ti % 5;
            const int sdx = (dxi - 2) * S;
            if ((sdx & 1) == 0) {
                const int off = rb[tr] + (sdx / 2) * 16;
                { const TwoPairs u = lds_quad_at(lds, off + 0 * C::PLANE); tp.l = u.a; tp.r = u.b; }
                { const TwoPairs u = lds_quad_at(lds, off + 1 * C::PLANE); tp.g = u.a; tp.v = u.b; }
                { const TwoPairs u = lds_quad_at(lds, off + 2 * C::PLANE); tp.nx = u.a; tp.ny = u.b; }
                { const TwoPairs u = lds_quad_at(lds, off + 3 * C::PLANE); tp.nz = u.a; tp.z = u.b; }
            } else {
                const int off = rb[tr] + ((sdx - 1) / 2) * 16 + 4;
                tp.l = lds_odd_pair(lds, off + 0 * C::PLANE);  tp.r = lds_odd_pair(lds, off + 0 * C::PLANE + 8);
                tp.g = lds_odd_pair(lds, off + 1 * C::PLANE);  tp.v = lds_odd_pair(lds, off + 1 * C::PLANE + 8);
                tp.nx = lds_odd_pair(lds, off + 2 * C::PLANE); tp.ny = lds_odd_pair(lds, off + 2 * C::PLANE + 8);
                tp.nz = lds_odd_pair(lds, off + 3 * C::PLANE); tp.z = lds_odd_pair(lds, off + 3 * C::PLANE + 8);
            }
        };
        auto taps = [&](auto zero_aware) {
            constexpr bool ZA = decltype(zero_aware)::value;
            auto weigh = [&](const int ti, const TapT<f2>& tp) {
                const int dx = ti / 5 - 2, dy = ti % 5 - 2;
                const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
                if (dx == 0 && dy == 0) {
                    const TwoPairs lr = lds_quad_at(lds, cb + 0 * C::PLANE), gv = lds_quad_at(lds, cb + 1 * C::PLANE);
                    tap_center<f2>(acc, lr.a, lr.b, gv.a, gv.b);
                    return;
                }
                f2 e0 = splat(kLogB3[adx] + kLogB3[ady], f2{});
                if (ZA) {
                    bool okA = rowv[dy + 2], okB = okA;
                    if (XE) {
                        okA = okA && xA + dx * S >= 0 && xA + dx * S < g.W;
                        okB = okB && xA + 1 + dx * S >= 0 && xA + 1 + dx * S < g.W;
                    }
                    e0 = f2{ okA ? e0.x : kNegInf, okB ? e0.y : kNegInf };
                }
                tap_eval<f2, ZA>(acc, k, tp, e0, len_class(adx, ady), sn2);
            };
            TapT<f2> t0, t1;
            load_tap(0, t0);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int ti = i < 12 ? i : i + 1, tn = i + 1 < 12 ? i + 1 : i + 2;
                if (i == 12) weigh(12, t0);
                if (i < 23) { if (i & 1) load_tap(tn, t0); else load_tap(tn, t1); }
                __builtin_amdgcn_sched_barrier(0);
                if (i & 1) weigh(ti, t1); else weigh(ti, t0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (any_zero) taps(std::true_type{}); else taps(std::false_type{});
        const TwoPairs flr = lds_quad_at(lds, cb + 0 * C::PLANE), fgv = lds_quad_at(lds, cb + 1 * C::PLANE);
        const float4 outA = finish2(acc.sw.x, acc.sl.x, acc.sr.x, acc.sg.x, acc.sv.x, flr.a.x, flr.b.x, fgv.a.x, fgv.b.x);
        const float4 outB = finish2(acc.sw.y, acc.sl.y, acc.sr.y, acc.sg.y, acc.sv.y, flr.a.y, flr.b.y, fgv.a.y, fgv.b.y);
        // every LDS read of this step has returned (its value was used): the slots may be recycled
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) flags[4 + wv] = t + 1;
        RMD_PHASE(1)
        store_aux();
#ifdef RMD_LC_ABL_NOOUT
        if (outA.x + outB.x == -12345.678f)
#endif
        if (jw >= jlo && jw < jhi) {
            float4* o = a.out + (size_t)(y - g.buf_row0) * (size_t)g.W + (size_t)xA;
            if (inA) o[0] = outA;
            if (inB) o[1] = outB;
        }
        RMD_PHASE(2)
    }
#ifdef RMD_ATROUS_TRACE
    if (lane == 0 && blockIdx.x < 8192 && (wv == 0 || wv == L::NC - 1)) {
        if (wv == 0) for (int i = 0; i < 3; ++i) g_atrous_phase[8 * blockIdx.x + i] = ph[i];
        else         for (int i = 0; i < 2; ++i) g_atrous_phase[8 * blockIdx.x + 3 + i] = ph[i];
    }
#endif
}

template <int S>
__global__ __launch_bounds__(LcCfg<S>::THREADS, 4) void atrous_lc_kernel(AtrousArgs a)
{
    using C = PairCfg<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char atrous_lds[];
    const int tid = threadIdx.x;
    const int pid = blockIdx.x;
    const int xcd = pid & (kXcds - 1), slot = pid >> 3;
    int L;
    if (slot < a.int_per_xcd) { L = xcd * a.int_per_xcd + slot; if (L >= a.total_int) return; }
    else { L = a.total_int + xcd * a.xe_per_xcd + (slot - a.int_per_xcd); if (L >= a.nblocks) return; }
    int strip, band, r, bh;
    if (L < a.total_int) {
        r = L % S; const int t = L / S;
        strip = a.xe_lo + t % a.n_int; band = t / a.n_int; bh = a.band_h;
    } else {
        const int Lx = L - a.total_int, nxe = a.nstrips - a.n_int;
        r = Lx % S; const int t = Lx / S, e = t % nxe;
        strip = e < a.xe_lo ? e : a.n_int + e; band = t / nxe; bh = a.band_h_xe;
    }
    const int x0 = strip * C::CW;
    const int yb = a.band_base + band * bh;
    const int lo = max(yb, a.row0), hi = min(yb + bh, a.row1);
    const int ybase = yb + r;
    const int jlo = lo > ybase ? (lo - ybase + S - 1) / S : 0;
    const int jhi = hi > ybase ? (hi - ybase + S - 1) / S : 0;
    if (jlo >= jhi) return;
    const bool xedge = (x0 - 2 * S < 0) || (x0 + C::CW + 2 * S > a.g.W);
#ifdef RMD_ATROUS_TRACE
    const unsigned long long trace_t0 = wall_clock64(), trace_c0 = __builtin_amdgcn_s_memtime();
#endif
    if (xedge) atrous_lc_body<S, true>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
    else       atrous_lc_body<S, false>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
#ifdef RMD_ATROUS_TRACE
    __syncthreads();
    if (tid == 0 && pid < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_atrous_trace[6 * pid + 0] = trace_t0;
        g_atrous_trace[6 * pid + 1] = wall_clock64();
        g_atrous_trace[6 * pid + 2] = ((unsigned long long)xcc << 32) | hw;
        g_atrous_trace[6 * pid + 3] = ((unsigned long long)L << 8) | (xedge ? 1u : 0u);
        g_atrous_trace[6 * pid + 4] = __builtin_amdgcn_s_memtime() - trace_c0;
        g_atrous_trace[6 * pid + 5] = (unsigned long long)(jhi - jlo);
    }
#endif
}

template <int S, int MODE = 0>                      // 0 pixel-pair (barriers), 1 loader/consumer, 2 quad (2 x 2 block per lane)
static int launch_pair(AtrousArgs a, hipStream_t stream)
{
    using C = PairCfg<S>;
    constexpr bool LC = MODE == 1, QD = MODE == 2;
    const void* fn = LC ? reinterpret_cast<const void*>(&atrous_lc_kernel<S>)
                   : QD ? reinterpret_cast<const void*>(&atrous_quad_kernel<S>) : reinterpret_cast<const void*>(&atrous_pair_kernel<S>);
    constexpr int lds_bytes = LC ? LcCfg<S>::LDS_BYTES : QD ? QuadCfg<S>::LDS_BYTES : C::LDS_BYTES;
    if (first_use_on_device(fn))
        RMD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    a.band_base = a.row0 / (8 * S) * (8 * S);
    const int rows = a.row1 - a.band_base;
    a.nstrips = (a.g.W + C::CW - 1) / C::CW;
    // strips whose staged columns leave the frame (same test as the kernel's)
    int xe_lo = 0, xe_hi = 0;
    for (int st = 0; st < a.nstrips && st * C::CW - 2 * S < 0; ++st) ++xe_lo;
    for (int st = a.nstrips - 1; st >= xe_lo && st * C::CW + C::CW + 2 * S > a.g.W; --st) ++xe_hi;
    a.xe_lo = xe_lo;
    a.n_int = a.nstrips - xe_lo - xe_hi;
    const int nxe = xe_lo + xe_hi;
    // Band count: every workgroup of a launch does the same work (border strips half of it, their body is
    // slower), so the launch runs in ceil(workgroups / resident slots) rounds; take the band count that
    // fills the rounds best, discounted by the 4 halo rows a workgroup stages on top of its own rows.
    // (3840 wide: 28 interior + 2 border strips = 32 S b workgroups, i.e. exactly 768 for S <= 8.)
    const int unit = S * (LC ? LcCfg<S>::NC : QD ? QuadCfg<S>::ADV : C::ADV);
    const int slots = (LC ? LcCfg<S>::WG_PER_CU : QD ? QuadCfg<S>::WG_PER_CU : C::WG_PER_CU) * a.cus;
    int best_nb = 1;
    double best = -1.0;
    for (int nb = 1; nb <= 64; ++nb) {
        int h = (rows + nb - 1) / nb;
        h = ((h + unit - 1) / unit) * unit;
        const int bands = (rows + h - 1) / h;
        int hx = ((h / 2 + unit - 1) / unit) * unit;
        const int bands_x = (rows + hx - 1) / hx;
        const int wgs = (a.n_int * bands + nxe * bands_x) * S;
        const int rounds = (wgs + slots - 1) / slots;
        const double lattice_rows = (double)h / S;
        const double eff = (double)wgs / ((double)rounds * slots) * lattice_rows / (lattice_rows + 4.0);
        if (eff > best + 1e-9) { best = eff; best_nb = nb; }
        if (h == unit) break;
    }
    int h = (rows + best_nb - 1) / best_nb;
    h = ((h + unit - 1) / unit) * unit;
    a.band_h = h;
    a.band_h_xe = ((h / 2 + unit - 1) / unit) * unit;
    const int nbands = (rows + h - 1) / h, nbands_x = (rows + a.band_h_xe - 1) / a.band_h_xe;
    a.total_int = a.n_int * nbands * S;
    a.nblocks = a.total_int + nxe * nbands_x * S;
    a.int_per_xcd = (a.total_int + kXcds - 1) / kXcds;
    a.xe_per_xcd = (a.nblocks - a.total_int + kXcds - 1) / kXcds;
    a.per_xcd = a.int_per_xcd + a.xe_per_xcd;
    if (LC)      hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_lc_kernel<S>), dim3(a.per_xcd * kXcds), dim3(LcCfg<S>::THREADS), lds_bytes, stream, a);
    else if (QD) hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_quad_kernel<S>), dim3(a.per_xcd * kXcds), dim3(256), lds_bytes, stream, a);
    else         hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_pair_kernel<S>), dim3(a.per_xcd * kXcds), dim3(256), lds_bytes, stream, a);
    RMD_LAUNCH_CHECK("atrous_pair_kernel");
    return RMD_OK;
}

static int launch_lc_iter(int iteration, const AtrousArgs& a, hipStream_t stream)
{
    switch (iteration) {
        case 0: return launch_pair<1, 1>(a, stream);
        case 1: return launch_pair<2, 1>(a, stream);
        case 2: return launch_pair<4, 1>(a, stream);
        case 3: return launch_pair<8, 1>(a, stream);
        default: return launch_pair<16, 1>(a, stream);
    }
}

static int launch_quad_iter(int iteration, const AtrousArgs& a, hipStream_t stream)
{
    switch (iteration) {
        case 0: return launch_pair<1, 2>(a, stream);
        case 1: return launch_pair<2, 2>(a, stream);
        case 2: return launch_pair<4, 2>(a, stream);
        case 3: return launch_pair<8, 2>(a, stream);
        default: return launch_pair<16, 2>(a, stream);
    }
}

static int launch_pair_iter(int iteration, const AtrousArgs& a, hipStream_t stream)
{
    switch (iteration) {
        case 0: return launch_pair<1>(a, stream);
        case 1: return launch_pair<2>(a, stream);
        case 2: return launch_pair<4>(a, stream);
        case 3: return launch_pair<8>(a, stream);
        default: return launch_pair<16>(a, stream);
    }
}

// Work decomposition of one stream launch: fills the band fields of `a`, returns the efficiency estimate
// (share of the resident workgroup slots used by whole rounds) x (own lattice rows / staged lattice rows).
template <int S, int NP>
static double plan_stream(AtrousArgs& a)
{
    using C = StreamCfg<S, NP>;
    // Bands are laid out from row0 rounded down to a multiple of 2S.  (They used to be aligned to
    // multiples of band_h in global rows: a strip that does not start on such a multiple then got one
    // band more than this heuristic planned, 780 workgroups for 768 slots, i.e. a second round --
    // interior ranks of a row-strip run paid 1.23 ms per frame instead of 0.96.)
    a.band_base = a.row0 / (2 * S) * (2 * S);
    const int rows = a.row1 - a.band_base;
    a.nstrips = (a.g.W + C::CW - 1) / C::CW;
    // one resident wave of workgroups: WG_PER_CU per CU (LDS-limited) x the CUs of the device
    const int per_band = a.nstrips * S;
    const int unit = S * C::ADV;                        // whole steps per lattice
    // Number of bands: every workgroup does the same work, so the launch runs in
    // ceil(workgroups / resident slots) rounds; pick the band count that fills the rounds best,
    // discounted by the 4 halo rows each workgroup stages on top of its own lattice rows.
    const int slots = C::WG_PER_CU * a.cus;
    // Candidates: nb bands for every strip, or nb + 1 for x of the strips, x as large as the rounds nb needs
    // anyway leave room for (3840 wide, step 2: 30 strips x 2 lattices x 12 bands = 720 workgroups for 768 slots
    // left 48 CUs with two workgroups instead of three; 24 strips with 13 bands make it 768).
    static const int mixed = [] { const char* e = getenv("RMD_ATROUS_MIXED_BANDS"); return e ? atoi(e) : 1; }();
    auto round_up = [&](int h) { return ((h + unit - 1) / unit) * unit; };
    int best_h = round_up(rows), best_hh = best_h, best_x = 0;
    double best = -1.0;
    for (int nb = 1; nb <= 64; ++nb) {
        const int h = round_up((rows + nb - 1) / nb);
        const int bands = (rows + h - 1) / h;
        const int wgs0 = bands * per_band;
        const int rounds = (wgs0 + slots - 1) / slots;
        int hh = h, x = 0;
        if (mixed && h > unit && rounds == 1) {          // (with two or more rounds the late workgroups backfill anyway)
            const int h2 = round_up((rows + bands) / (bands + 1));
            const int bands2 = (rows + h2 - 1) / h2;
            // only where the two sizes stay close (>= 6 bands): a CU runs whichever three workgroups it is
            // dealt, and 4 against 3 bands (let alone 2 against 1) leaves the CUs with the big ones behind
            if (h2 < h && bands2 > bands && h2 * 100 >= h * 85) {
                x = min(a.nstrips, (rounds * slots - wgs0) / (S * (bands2 - bands)));
                if (x > 0) hh = h2;
            }
        }
        const int bands_hi = (rows + hh - 1) / hh;
        const int wgs = wgs0 + x * S * (bands_hi - bands);
        // staged lattice rows of the launch over what its rounds could hold; a CU's time is the sum of its workgroups
        const double avg_rows = ((double)(a.nstrips - x) * bands * h + (double)x * bands_hi * hh) / ((double)(a.nstrips - x) * bands + (double)x * bands_hi) / S;
        const double eff = (double)wgs / ((double)rounds * slots) * avg_rows / (avg_rows + 4.0);
        if (eff > best + 1e-9) { best = eff; best_h = h; best_hh = hh; best_x = x; }
        if (h == unit) break;
    }
    a.band_h = best_h;
    a.band_h_hi = best_hh;
    a.n_hi = best_x;
    const int nbands = (rows + best_h - 1) / best_h, nbands_hi = (rows + best_hh - 1) / best_hh;
    a.nblocks_hi = best_x * S * nbands_hi;
    a.nblocks = a.nblocks_hi + (a.nstrips - best_x) * S * nbands;
    a.per_xcd = (a.nblocks + kXcds - 1) / kXcds;
    return best;
}

template <int S, int NP>
static int launch_planned(const AtrousArgs& a, hipStream_t stream)
{
    using C = StreamCfg<S, NP>;
    // per device, not per process: a host that drives several GPUs through rmd_set_device needs the
    // attribute (NP = 1 at step 16 asks for 65 568 bytes of LDS) on every one of them
    if (first_use_on_device(reinterpret_cast<const void*>(&atrous_stream_kernel<S, NP>)))
        RMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&atrous_stream_kernel<S, NP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_stream_kernel<S, NP>), dim3(a.per_xcd * kXcds), dim3(256), C::LDS_BYTES, stream, a);
    RMD_LAUNCH_CHECK("atrous_stream_kernel");
    return RMD_OK;
}

template <int S, int NP>
static int launch_stream(AtrousArgs a, hipStream_t stream)
{
    plan_stream<S, NP>(a);
    return launch_planned<S, NP>(a, stream);
}

// Library default: 128-column strips with two row pairs per step (NP = 2).  The other two decompositions were
// measured against it with bench.py's timing and never won: four row pairs on 64-column strips (NP = 4: 127 /
// 132 / 131 / 139 / 142 us per 4K iteration against 125 / 131 / 132 / 133 / 140) and one row pair on 256-column
// strips (NP = 1, two workgroups per CU: 84 us per iteration on a 7680x540 strip against 74-80, and 84.5 against
// 86.4 only at step 16, where NP = 2 has 960 workgroups for 768 slots).  RMD_ATROUS_NP = 1 | 4 forces them.
template <int S>
static int launch_stream_auto(AtrousArgs a, hipStream_t stream)
{
    static const int force = [] { const char* e = getenv("RMD_ATROUS_NP"); return e ? atoi(e) : 0; }();
    if (force == 1) return launch_stream<S, 1>(a, stream);
    if (force == 4 && S <= 8) return launch_stream<S, 4>(a, stream);
    return launch_stream<S, 2>(a, stream);
}

template <int NP>
static int launch_stream_iter(int iteration, const AtrousArgs& a, hipStream_t stream)
{
    switch (iteration) {
        case 0: return launch_stream<1, NP>(a, stream);
        case 1: return launch_stream<2, NP>(a, stream);
        case 2: return launch_stream<4, NP>(a, stream);
        case 3: return launch_stream<8, NP>(a, stream);
        default: return launch_stream<16, NP>(a, stream);
    }
}

static int launch_stream_auto_iter(int iteration, const AtrousArgs& a, hipStream_t stream)
{
    switch (iteration) {
        case 0: return launch_stream_auto<1>(a, stream);
        case 1: return launch_stream_auto<2>(a, stream);
        case 2: return launch_stream_auto<4>(a, stream);
        case 3: return launch_stream_auto<8>(a, stream);
        default: return launch_stream_auto<16>(a, stream);
    }
}

}  // namespace rmd

using namespace rmd;

extern "C" int rmd_debug_atrous_plan(int width, int height, int row0, int row1, int iteration, int cus, int* out)
{
    if (!out) return fail(RMD_E_NULL, "rmd_debug_atrous_plan: out is NULL");
    if (width < 1 || height < 1 || row0 < 0 || row1 > height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_debug_atrous_plan: frame %dx%d rows [%d,%d) invalid", width, height, row0, row1);
    if (iteration < 0 || iteration > 4) return fail(RMD_E_PARAM, "rmd_debug_atrous_plan: iteration %d outside [0,4]", iteration);
    if (cus < 1) return fail(RMD_E_PARAM, "rmd_debug_atrous_plan: cus %d", cus);
    AtrousArgs a;
    a.g = Geom{ width, height, 0, height };
    a.row0 = row0; a.row1 = row1; a.step = 1 << iteration; a.cus = cus;
    a.n_hi = a.band_h_hi = a.nblocks_hi = 0;
    switch (iteration) {
        case 0: plan_stream<1, 2>(a); break;
        case 1: plan_stream<2, 2>(a); break;
        case 2: plan_stream<4, 2>(a); break;
        case 3: plan_stream<8, 2>(a); break;
        default: plan_stream<16, 2>(a); break;
    }
    out[0] = a.nblocks; out[1] = a.nstrips; out[2] = a.band_base; out[3] = a.band_h; out[4] = a.band_h_hi;
    out[5] = a.n_hi; out[6] = a.nblocks_hi; out[7] = a.per_xcd;
    return RMD_OK;
}

extern "C" int rmd_debug_atrous_protocol_errors(unsigned int* count)
{
    if (!count) return fail(RMD_E_NULL, "rmd_debug_atrous_protocol_errors: count is NULL");
    RMD_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_atrous_lc_errors), sizeof(unsigned int)));
    return RMD_OK;
}

extern "C" int rmd_svgf_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                               const float* in, float* out, int row0, int row1, void* stream)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_atrous: params is NULL");
    if (!in || !out || !f->nd) return fail(RMD_E_NULL, "rmd_svgf_atrous: in/out/nd plane is NULL");
    if (in == out) return fail(RMD_E_BUFFER, "rmd_svgf_atrous: in and out alias (taps cross pixels)");
    if (iteration < 0 || iteration > 12) return fail(RMD_E_PARAM, "rmd_svgf_atrous: iteration %d outside [0,12]", iteration);
    if (!(p->sigma_n > 0.0f) || !(p->sigma_z > 0.0f) || !(p->sigma_l > 0.0f))
        return fail(RMD_E_PARAM, "rmd_svgf_atrous: sigma_n/sigma_z/sigma_l must be > 0");
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_atrous: rows [%d,%d) invalid", row0, row1);
    const int s = 1 << iteration;
    if (int e = check_rows_in_buffer(f, row0 - 2 * s, row1 + 2 * s, "rmd_svgf_atrous")) return e;
    if (!aligned_to(in, 16) || !aligned_to(out, 16) || !aligned_to(f->nd, 16))
        return fail(RMD_E_ALIGN, "rmd_svgf_atrous: float4 planes must be 16-byte aligned");

    AtrousArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.in = (const float4*)in; a.nd = (const float4*)f->nd; a.out = (float4*)out;
    a.row0 = row0; a.row1 = row1; a.step = s;
    a.sigma_n = p->sigma_n; a.sigma_z = p->sigma_z; a.sigma_l = p->sigma_l;
    a.band_h = a.band_base = a.nstrips = a.nblocks = a.per_xcd = 0;
    a.n_int = a.xe_lo = a.band_h_xe = a.total_int = a.int_per_xcd = a.xe_per_xcd = 0;
    a.n_hi = a.band_h_hi = a.nblocks_hi = 0;
    a.nt_out = (double)(row1 - row0) * f->width * 48.0 > 256.0e6 ? 1 : 0;
    if (p->atrous_cus < 0) return fail(RMD_E_PARAM, "rmd_svgf_atrous: atrous_cus %d is negative", p->atrous_cus);
    a.cus = p->atrous_cus > 0 && p->atrous_cus < device_cus() ? p->atrous_cus : device_cus();

    int variant = p->atrous_variant;
    if (variant == 0) {
        if (iteration <= 4) return launch_stream_auto_iter(iteration, a, as_stream(stream));
        variant = 1;
    }
    if ((variant == 2 || variant == 3) && iteration > 4)
        return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
    if (variant == 4) {
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        return launch_pair_iter(iteration, a, as_stream(stream));                       // pixel pairs, SoA LDS planes
    }
    if (variant == 7) {
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        return launch_lc_iter(iteration, a, as_stream(stream));                         // pixel pairs, loader wave + 4 compute waves
    }
    if (variant == 8) {
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        return launch_quad_iter(iteration, a, as_stream(stream));                       // 2 x 2 pixel block per lane
    }
    if (variant == 5) {
        dim3 grid5((f->width + 63) / 64, (row1 - row0 + 3) / 4);
        hipLaunchKernelGGL(atrous_direct2_kernel, grid5, dim3(256), 0, as_stream(stream), a);
        RMD_LAUNCH_CHECK("atrous_direct2_kernel");
        return RMD_OK;
    }
    if (variant == 6) {                                                               // 64 columns x 4 row pairs
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        return launch_stream_iter<4>(iteration, a, as_stream(stream));
    }
    if (variant == 2) return launch_stream_iter<1>(iteration, a, as_stream(stream));   // 256 columns x 1 row pair
    if (variant == 3) return launch_stream_iter<2>(iteration, a, as_stream(stream));   // 128 columns x 2 row pairs
    if (variant != 1) return fail(RMD_E_PARAM, "rmd_svgf_atrous: unknown atrous_variant %d", p->atrous_variant);
    dim3 grid((f->width + 63) / 64, (row1 - row0 + 3) / 4);
    hipLaunchKernelGGL(atrous_direct_kernel, grid, dim3(256), 0, as_stream(stream), a);
    RMD_LAUNCH_CHECK("atrous_direct_kernel");
    return RMD_OK;
}
