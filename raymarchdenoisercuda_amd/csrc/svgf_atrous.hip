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
//            conflict-free ds_read_b128 with an immediate offset (the per-pixel setup's ds_read_b32
//            of the neighbours' .w at a 16-byte stride are not: 15 % of the LDS cycles are bank
//            conflicts, profiles/r03_atrous_pmc.txt; LDS is not what bounds the pass).  Each thread produces two
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
#include "svgf_tv.h"
#include "pixel_convert.h"
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
    // a SECOND row range in the same launch (stream kernel; the two boundary bands of a strip's exchanged iteration, which would
    // otherwise be two launches of one step each): its own band plan, workgroups nblocks .. nblocks + b_nblocks - 1
    int b_row0, b_row1, b_band_h, b_band_base, b_nblocks, b_n_hi, b_band_h_hi, b_nblocks_hi;
    int cus;       // CUs the launch may count on (rmd_svgf_params.atrous_cus or the whole device)
    int nt_out;    // store the outputs non-temporally (launches whose planes overflow the 256 MB Infinity Cache)
    // 8-bit back end (the LAST iteration of rmd_svgf_gbuffer_frame, the OUT8 instantiations): the result is multiplied by the
    // GBuffer's albedo, quantised as rmd_convert_f32_to_u8 does (pixel_convert.h) and stored as uchar4 to `out8` -- 4 B/px
    // written instead of 16, and no conversion launch behind the frame; `out` is not written.
    const uchar4* albedo8; uchar4* out8;
#ifdef RMD_EXPERIMENTS
    // SIDE JOB (stream kernel, rmd_svgf_frame_atrous_next; measured and lost): the temporal pass of the
    // NEXT frame, one 64x4 tile per workgroup every side_every steps, tiles claimed from a device counter (units 0 ..
    // side_units-1 = the tiles of `side`'s rows, row major).  side_units = 0: none.
    TemporalArgs side;
    unsigned* side_counter;
    int side_units, side_every;
#endif
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

// Appendix A.A.2 without a branch: both normals zero => w_n = 1, exactly one zero => 0.  With zc = 1 for a zero centre
// normal (else 0) and ft = 1 for a zero tap normal (else 0), the cosine becomes clamp01(n_p.n_t + zc * ft): the dot product
// itself (same bits) for zc = 0, where a zero tap gives 0 and log2(0) = -inf; 1 for two zero normals; 0 for a zero centre
// and a non-zero tap.  ft costs v_max3_f32(|x|,|y|,|z|) + compare + select per tap, the rest stays packed.
__device__ __forceinline__ float tap_is_zero(const float4 n)
{
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z)) == 0.0f ? 1.0f : 0.0f;
}
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
    float e = fma_(sigma_n, log2_(cosine), e0);      // (zero-aware callers pass the cosine of tap_is_zero's rule)
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
    float cosine;
    if (ZERO_AWARE) {
        float d = k.nx * t.n.x;
        d = fma_(k.ny, t.n.y, d);
        d = fma_(k.nz, t.n.z, d);
        cosine = clamp01(fma_(x.zero ? 1.0f : 0.0f, tap_is_zero(t.n), d));
    } else {
        cosine = tap_cosine<float>(k, t);
    }
    const float e = tap_exponent<ZERO_AWARE>(cosine, e0, sigma_n, k.z, k.lum, k.il, x, t, false, adx, ady);
    tap_accumulate<float>(s, exp2_(e), t);
}

// the (A,B) pair sharing one tap: adyA / adyB are the tap's |row offset| seen from A and from B.
// Cosine and accumulation are packed (v_pk_*), the exponents scalar; the empty asm keeps the
// optimizer from re-vectorising the exponents (it would gather iz[] pairs through scratch).
template <bool ZERO_AWARE>
__device__ __forceinline__ void tap_pair(Acc<f2>& s, const Center<f2>& k, const CenterAux& xa, const CenterAux& xb,
                                         const Tap& t, float e0A, float e0B, int adx, int adyA, int adyB, float sigma_n)
{
    f2 c;
    if constexpr (ZERO_AWARE) {
        f2 d = k.nx * f2{ t.n.x, t.n.x };
        d = fma_(k.ny, f2{ t.n.y, t.n.y }, d);
        d = fma_(k.nz, f2{ t.n.z, t.n.z }, d);
        const float ft = tap_is_zero(t.n);
        c = pk_fma_lo_clamp(f2{ xa.zero ? 1.0f : 0.0f, xb.zero ? 1.0f : 0.0f }, f2{ ft, ft }, d);
    } else {
        c = tap_cosine<f2>(k, t);
    }
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
}

// The wave whose centres ALL have a zero normal (the background of the Cornell planes: 65 % of their pixels): Appendix A.A.2 leaves
// w_n = 1 for a zero tap normal and 0 for any other, so there is no cosine and no logarithm to take -- the zero-aware form
// computes clamp01(0 . n_t + 1 * ft) = ft and then sigma_n * log2(ft) + e0, which is e0 exactly for ft = 1 (log2 1 = 0, and
// fma(sigma_n, 0, e0) rounds nothing) and -inf for ft = 0.  Same bits, without 5 packed + 1 plain + 2 transcendental
// instructions of the 27 slot-equivalents a zero-aware pair tap costs.
__device__ __forceinline__ void tap_pair_zero_centres(Acc<f2>& s, const Center<f2>& k, const CenterAux& xa, const CenterAux& xb,
                                                      const Tap& t, float e0A, float e0B, int adx, int adyA, int adyB)
{
    const bool tz = tap_is_zero(t.n) != 0.0f;
    f2 e = f2{ tz ? e0A : kNegInf, tz ? e0B : kNegInf };
    const f2 dz = k.z - f2{ t.n.w, t.n.w }, dl = k.lum - f2{ t.c.x, t.c.x };
    if (adx | adyA) e.x = fma_(-fabsf(dz.x), xa.iz[len_class(adx, adyA)], e.x);
    if (adx | adyB) e.y = fma_(-fabsf(dz.y), xb.iz[len_class(adx, adyB)], e.y);
    e.x = fma_(-fabsf(dl.x), k.il.x, e.x);
    e.y = fma_(-fabsf(dl.y), k.il.y, e.y);
    tap_accumulate<f2>(s, exp2_(e), t);
}
__device__ __forceinline__ void tap_single_zero_centre(Acc<float>& s, const Center<float>& k, const CenterAux& x, const Tap& t,
                                                       float e0, int adx, int ady)
{
    float e = tap_is_zero(t.n) != 0.0f ? e0 : kNegInf;
    if (adx | ady) e = fma_(-fabsf(k.z - t.n.w), x.iz[len_class(adx, ady)], e);
    e = fma_(-fabsf(k.lum - t.c.x), k.il, e);
    tap_accumulate<float>(s, exp2_(e), t);
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
    const float4 res = finish(lone.sw + acc.sw, lone.sl + acc.sl, lone.sr + acc.sr, lone.sg + acc.sg, lone.sv + acc.sv, ctr.c);
    if (a.out8) a.out8[i] = u8_from_float4(res, true, float4_from_u8(a.albedo8[i], false, 0.0f));
    else a.out[i] = res;
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
template <int S, int NP, bool EDGE, bool OUT8>
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
    unsigned pa8[2] = { 0u, 0u };                     // OUT8: the albedo bytes of the two output pixels of the current step

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
    constexpr int kPieces = 4 + C::NT + 7 + (OUT8 ? 2 : 0);
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
        } else if (i == 4 + C::NT + 6) {
            // halo columns x0-1 and x0+CW of the 4*NP variance rows: lanes 0 .. 8NP-1 keep theirs
            const int t8 = tid & (8 * NP - 1);
            const int ii = t8 >> 2, ud = (t8 >> 1) & 1, side = t8 & 1;
            const int yy = clamp_row(ybase + (jo + ii) * S + (ud ? 1 : -1));
            const int xx = side ? x0 + C::CW : x0 - 1;
            pvh = in_f[((size_t)(yy - g.buf_row0) * (size_t)g.W + (size_t)xx) * 4 + 3];
        } else if constexpr (OUT8) {
            // the albedo of THIS step's two output pixels (jo - ADV = the step being computed): consumed by write_out
            const int r = i - (4 + C::NT + 7);
            const int y = clamp_row(ybase + (jo - C::ADV + 2 * pr + r) * S);
            pa8[r] = (reinterpret_cast<const unsigned*>(a.albedo8) + row_base(y, x0))[col];
        }
    };
    // OUT8, frame-edge form: the same two fetches at the top of the step, per-lane tested
    auto load_albedo = [&](const int j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int jj = j + 2 * pr + i;
            pa8[i] = (jj >= jlo && jj < jhi && xin) ? (reinterpret_cast<const unsigned*>(a.albedo8) + row_base(ybase + jj * S, x0))[col] : 0u;
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

    // The same for both pixels of the pair at once, interior form (no border tests): the prefilter's nine terms, the depth
    // differences and the six reciprocals' arguments as packed f32 on the (A, B) register pairs -- the operations of setup() /
    // make_center() lane for lane, so the same bits -- 19 instructions fewer per step.
    auto setup_pair = [&](const int rbA, const int rbB, const Tap& tA, const Tap& tB, Center<f2>& k, CenterAux& xA, CenterAux& xB) {
        const int cA = rbA + 2 * S * 16, cB = rbB + 2 * S * 16;
        const float* vuA = var_lds + ((2 * pr + 0) * 2 + 0) * C::VAR_ROW + col;
        const float* vdA = var_lds + ((2 * pr + 0) * 2 + 1) * C::VAR_ROW + col;
        const float* vuB = var_lds + ((2 * pr + 1) * 2 + 0) * C::VAR_ROW + col;
        const float* vdB = var_lds + ((2 * pr + 1) * 2 + 1) * C::VAR_ROW + col;
        // prefilter9(ul, l, dl, u, c, d, ur, r, dr)
        f2 v = pair_of(vuA[0], vuB[0]) * f2{ 0.0625f, 0.0625f };
        v = fma_(f2{ 0.125f, 0.125f }, pair_of(lds_f1(lds, cA - 16 + 12), lds_f1(lds, cB - 16 + 12)), v);
        v = fma_(f2{ 0.0625f, 0.0625f }, pair_of(vdA[0], vdB[0]), v);
        v = fma_(f2{ 0.125f, 0.125f }, pair_of(vuA[1], vuB[1]), v);
        v = fma_(f2{ 0.25f, 0.25f }, pair_of(tA.c.w, tB.c.w), v);
        v = fma_(f2{ 0.125f, 0.125f }, pair_of(vdA[1], vdB[1]), v);
        v = fma_(f2{ 0.0625f, 0.0625f }, pair_of(vuA[2], vuB[2]), v);
        v = fma_(f2{ 0.125f, 0.125f }, pair_of(lds_f1(lds, cA + 16 + 12), lds_f1(lds, cB + 16 + 12)), v);
        v = fma_(f2{ 0.0625f, 0.0625f }, pair_of(vdA[2], vdB[2]), v);
        const f2 z = pair_of(tA.n.w, tB.n.w);
        const f2 dzr = pair_of(lds_f1(lds, C::PLANE_BYTES + cA + 16 + 12), lds_f1(lds, C::PLANE_BYTES + cB + 16 + 12)) - z;
        const f2 dzd = pair_of(zd_cur[0], zd_cur[1]) - z;
        const float gzA = fabsf(dzr.x) + fabsf(dzd.x), gzB = fabsf(dzr.y) + fabsf(dzd.y);
        // make_center
        k.nx = pair_of(tA.n.x, tB.n.x); k.ny = pair_of(tA.n.y, tB.n.y); k.nz = pair_of(tA.n.z, tB.n.z); k.z = z;
        k.lum = pair_of(tA.c.x, tB.c.x);
        xA.zero = is_zero3(tA.n); xB.zero = is_zero3(tB.n);
        constexpr float kInvLog2e = 1.0f / kLog2e, kEps = 1e-8f / kLog2e;
        const f2 sd = { __builtin_amdgcn_sqrtf(v.x > 0.0f ? v.x : 0.0f), __builtin_amdgcn_sqrtf(v.y > 0.0f ? v.y : 0.0f) };
        const f2 ild = fma_(f2{ a.sigma_l * kInvLog2e, a.sigma_l * kInvLog2e }, sd, f2{ kEps, kEps });
        k.il = f2{ fast_rcp(ild.x), fast_rcp(ild.y) };
        const f2 za = f2{ a.sigma_z * fmaxf(gzA, 1e-8f), a.sigma_z * fmaxf(gzB, 1e-8f) } * f2{ (float)S, (float)S };
        constexpr float kLen[5] = { 1.0f, 1.41421356237309504880f, 2.0f, 2.23606797749978969641f, 2.82842712474619009760f };
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const f2 den = fma_(za, f2{ kLen[c] * kInvLog2e, kLen[c] * kInvLog2e }, f2{ kEps, kEps });
            xA.iz[c] = fast_rcp(den.x); xB.iz[c] = fast_rcp(den.y);
        }
    };

    // ---- one step: this thread's outputs are lattice rows jw (A) and jw+1 (B), jw = j + 2*pr
    auto compute = [&](const int j) {
        const int jw = j + 2 * pr;
        // Window rows jw-2 .. jw+3 sit in ring slots slot_of(jw-2) + tr (mod NR).  jw is even and NR is even, so the rows
        // come in three pairs that never straddle the wrap: one address register per PAIR, the odd row of a pair is an
        // immediate offset (3 VGPRs instead of 6 across the tap loop).
        int rbp[3];
        {
            int slot = slot_of(jw - 2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                rbp[q] = slot * C::ROW_BYTES + col * 16;
                slot = slot >= C::NR - 2 ? slot + 2 - C::NR : slot + 2;
            }
        }
        auto rb = [&](const int tr) { return rbp[tr >> 1] + (tr & 1) * C::ROW_BYTES; };
        const int yA = ybase + jw * S, yB = yA + S;
        Tap cA, cB;
        cA.c = lds_f4(lds, rb(2) + 2 * S * 16); cA.n = lds_f4(lds, C::PLANE_BYTES + rb(2) + 2 * S * 16);
        cB.c = lds_f4(lds, rb(3) + 2 * S * 16); cB.n = lds_f4(lds, C::PLANE_BYTES + rb(3) + 2 * S * 16);
        Center<float> kA, kB;
        CenterAux xA, xB;
        Center<f2> kAB;
        if constexpr (EDGE) {
            setup(0, yA, rb(2), cA, kA, xA);
            setup(1, yB, rb(3), cB, kB, xB);
            kAB = pack(kA, kB);
        } else {
            setup_pair(rb(2), rb(3), cA, cB, kAB, xA, xB);
        }
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
        const bool all_zero = __builtin_amdgcn_ballot_w64(!(xA.zero && xB.zero)) == 0ull;       // every centre of the wave: no cosine to take
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
                const int off = rb(tr0 + q) + (2 * S + (dxi - 2) * S) * 16;
                t[q].c = lds_f4(lds, off);
                t[q].n = lds_f4(lds, C::PLANE_BYTES + off);
            }
        };
        auto taps = [&](auto zero_mode) {             // 0: no centre of the wave has a zero normal | 1: some have | 2: all have
            constexpr int ZM = decltype(zero_mode)::value;
            constexpr bool ZA = ZM != 0;
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
                    if constexpr (ZM == 2) {
                        if (tr == 0)      tap_single_zero_centre(sA, kA, xA, t[q], e0A, adx, adyA);
                        else if (tr == 5) tap_single_zero_centre(sB, kB, xB, t[q], e0B, adx, adyB);
                        else              tap_pair_zero_centres(sAB, kAB, xA, xB, t[q], e0A, e0B, adx, adyA, adyB);
                    } else {
                        if (tr == 0)      tap_single<ZA>(sA, kA, xA, t[q], e0A, adx, adyA, sn);
                        else if (tr == 5) tap_single<ZA>(sB, kB, xB, t[q], e0B, adx, adyB, sn);
                        else              tap_pair<ZA>(sAB, kAB, xA, xB, t[q], e0A, e0B, adx, adyA, adyB, sn);
                    }
                }
            };
            Tap t0[GR], t1[GR];
            load_grp(0, t0);
            static_assert(kPieces <= NG || EDGE, "one refill instruction per tap group");
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
#ifndef RMD_PIECES_PER_GROUP
#define RMD_PIECES_PER_GROUP 1
#endif
                if (!EDGE) {
#pragma unroll
                    for (int q = 0; q < RMD_PIECES_PER_GROUP; ++q)
                        if (grp * RMD_PIECES_PER_GROUP + q < kPieces) prefetch_piece(grp * RMD_PIECES_PER_GROUP + q, j - 2 + C::NR, j + C::ADV);
                }
                if (grp & 1) { if (grp < NG - 1) load_grp(grp + 1, t0); weigh_grp(grp, t1); }
                else         { if (grp < NG - 1) load_grp(grp + 1, t1); weigh_grp(grp, t0); }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (all_zero)      taps(std::integral_constant<int, 2>{});
        else if (any_zero) taps(std::integral_constant<int, 1>{});
        else               taps(std::integral_constant<int, 0>{});

        // Per-pixel sums = (row only this pixel taps: dy=-2 for role A, dy=+2 for role B) + (the four
        // rows shared with its partner), each summed dx outer / dy inner.  The direct kernel groups
        // its 25 taps the same way, so the two variants stay bit-identical.
        // The centre colours are only needed again for the pass-through case of finish(): re-read from the ring (through an
        // address the optimizer cannot prove equal, or it keeps the first copy) instead of held in 8 VGPRs across the tap loop.
        int ra = rb(2), rbb = rb(3);
        asm volatile("" : "+v"(ra), "+v"(rbb));
        const float4 ccA = lds_f4(lds, ra + 2 * S * 16), ccB = lds_f4(lds, rbb + 2 * S * 16);
        outA = finish(sA.sw + sAB.sw.x, sA.sl + sAB.sl.x, sA.sr + sAB.sr.x, sA.sg + sAB.sg.x, sA.sv + sAB.sv.x, ccA);
        outB = finish(sB.sw + sAB.sw.y, sB.sl + sAB.sl.y, sB.sr + sAB.sr.y, sB.sg + sAB.sg.y, sB.sv + sAB.sv.y, ccB);
    };
    // The two output pixels of step j are written AFTER the ring refill of the step: the refill has to
    // wait for the prefetch loads (vmcnt), and stores issued before it would be waited for as well.
    auto write_out = [&](const int j) {
        const int jw = j + 2 * pr;
        const int yA = ybase + jw * S;
        if constexpr (OUT8) {
            // modulate by the albedo, quantise, store bytes: rmd_convert_f32_to_u8's arithmetic on the registers of finish()
            if (xin) {
                unsigned bA, bB;
                u8_pair_modulated(outA, outB, pa8[0], pa8[1], bA, bB);          // both pixels at once, packed (pixel_convert.h)
                unsigned* const o32 = reinterpret_cast<unsigned*>(a.out8);
                if (jw >= jlo && jw < jhi) (o32 + row_base(yA, x0))[col] = bA;
                if (jw + 1 >= jlo && jw + 1 < jhi) (o32 + row_base(yA + S, x0))[col] = bB;
            }
        } else if (xin) {
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
#ifdef RMD_EXPERIMENTS
    // Side job (experiments build): every side_every-th step the workgroup also runs one 64x4 tile of the NEXT frame's
    // temporal pass (thread = pixel, temporal_tile: the standalone pass's code and bits).  That pass is HBM-bound with a half
    // idle VALU, this one VALU-bound with HBM at 40 %; as waves of their own they find no registers beside three a-trous
    // workgroups, as a side job of these waves they need none: between two steps only the loop state is live (+10 VGPRs).
    // The tile is claimed (one atomic by thread 0) at the top of the step and its number handed to the workgroup through LDS
    // behind the step's second barrier, so nobody waits for the atomic.  MEASURED AND LOST: the four
    // waves of the workgroup sit through three dependent memory round trips per tile, and the two workgroups left on the CU
    // cannot fill its VALUs meanwhile: a tile costs 9.9 ns of frame time here against 5.7 ns in the pass's own launch.
    unsigned* const side_slot = reinterpret_cast<unsigned*>(lds + C::LDS_BYTES);
    const bool side_on = a.side_units > 0;
    int side_phase = side_on ? (int)(blockIdx.x % (unsigned)a.side_every) : 0;     // workgroups of a CU take their turns apart
    auto side_tile = [&](const unsigned u) {
        if (u < (unsigned)a.side_units) {
            const int tx = (int)(u % (unsigned)a.side.tiles_x), ty = (int)(u / (unsigned)a.side.tiles_x);
            temporal_tile(a.side, tx, a.side.row0 / 4 + ty);
        }
    };
#endif
    for (int j = j0; j < jhi; j += C::ADV) {
        if (j >= jq3)      __builtin_amdgcn_s_setprio(0);
        else if (j >= jq2) __builtin_amdgcn_s_setprio(1);
        else if (j >= jq1) __builtin_amdgcn_s_setprio(2);
        const bool more = j + C::ADV < jhi;
#ifdef RMD_EXPERIMENTS
        const bool side_now = side_on && side_phase == 0;                          // workgroup-uniform
        side_phase = side_phase == 0 ? a.side_every - 1 : side_phase - 1;
        unsigned claim = 0u;
        if (side_now && tid == 0) claim = atomicAdd(a.side_counter, 1u);
#endif
        if (EDGE && more) { load_rows(j - 2 + C::NR); load_aux(j + C::ADV); }     // in flight during compute (interior form: inside compute)
        if (EDGE && OUT8) load_albedo(j);
        RMD_PHASE(0)
        compute(j);
        RMD_PHASE(1)
        if (!more) {
            write_out(j);
#ifdef RMD_EXPERIMENTS
            if (side_now) {
                if (tid == 0) *side_slot = claim;
                __syncthreads();
                side_tile(*side_slot);
            }
#endif
            break;
        }
        __syncthreads();                                     // every wave is done reading the ADV oldest rows
        RMD_PHASE(2)
        store_rows(j - 2 + C::NR); store_aux();
        write_out(j);
#ifdef RMD_EXPERIMENTS
        if (side_now && tid == 0) *side_slot = claim;
#endif
        RMD_PHASE(3)
        __syncthreads();
        RMD_PHASE(4)
#ifdef RMD_EXPERIMENTS
        if (side_now) side_tile(*side_slot);                 // (ends with a barrier of its own when the pass keeps tile flags)
#endif
    }
#ifdef RMD_ATROUS_TRACE
    if (tid == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 5; ++i) g_atrous_phase[8 * blockIdx.x + i] = ph[i];
#endif
}


template <int S, int NP, bool OUT8>
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
    int L = (pid & (kXcds - 1)) * a.per_xcd + (pid >> 3);
    if (L >= a.nblocks + a.b_nblocks) return;
    // the launch's second row range, if any, follows the first in the workgroup order (scalar selects: L is wave-uniform)
    const bool second = L >= a.nblocks;
    if (second) L -= a.nblocks;
    const int p_row0 = second ? a.b_row0 : a.row0, p_row1 = second ? a.b_row1 : a.row1;
    const int p_band_h = second ? a.b_band_h : a.band_h, p_band_base = second ? a.b_band_base : a.band_base;
    const int p_n_hi = second ? a.b_n_hi : a.n_hi, p_band_h_hi = second ? a.b_band_h_hi : a.band_h_hi;
    const int p_nblocks_hi = second ? a.b_nblocks_hi : a.nblocks_hi;
    // Two groups of strips: n_hi strips (the outermost ones: their frame-edge body is the slower one) are cut into
    // one band more than the others, so that the workgroup count lands on the resident slots (plan_stream).
    int strip, band, bh;
    const int r = L % S;
    if (L < p_nblocks_hi) {
        const int t = L / S, e = t % p_n_hi;
        strip = e < p_n_hi / 2 ? e : a.nstrips - (p_n_hi - e); band = t / p_n_hi; bh = p_band_h_hi;
    } else {
        const int t = (L - p_nblocks_hi) / S, n_lo = a.nstrips - p_n_hi;
        strip = p_n_hi / 2 + t % n_lo; band = t / n_lo; bh = p_band_h;
    }
    const int x0 = strip * C::CW;
    // bands start at band_base + k*bh; band_base is row0 rounded down to a multiple of 2S, which is
    // all the (A,B) pairing needs (global lattice index floor(y/S) even at the top of a band)
    const int yb = p_band_base + band * bh;
    const int lo = max(yb, p_row0), hi = min(yb + bh, p_row1);
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
    if (edge) atrous_stream_body<S, NP, true, OUT8>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
    else      atrous_stream_body<S, NP, false, OUT8>(a, atrous_lds, tid, x0, ybase, jlo, jhi);
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
        g_atrous_trace[6 * pid + 5] = (unsigned long long)(jhi - jlo) | ((unsigned long long)S << 32);   // lattice rows it produced | the launch's step
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



#ifdef RMD_EXPERIMENTS
#include "experiments/atrous_variants.inc"
#endif

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
    static const int mixed = tuning_env("RMD_ATROUS_MIXED_BANDS", 1);
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

template <int S, int NP, bool OUT8>
static int launch_planned_as(const AtrousArgs& a, hipStream_t stream)
{
    using C = StreamCfg<S, NP>;
    // per device, not per process: a host that drives several GPUs through rmd_set_device needs the
    // attribute (NP = 1 at step 16 asks for 65 568 bytes of LDS) on every one of them
    if (first_use_on_device(reinterpret_cast<const void*>(&atrous_stream_kernel<S, NP, OUT8>)))
        RMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&atrous_stream_kernel<S, NP, OUT8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + 16));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_stream_kernel<S, NP, OUT8>), dim3(a.per_xcd * kXcds), dim3(256), C::LDS_BYTES + 16, stream, a);   // + the side job's slot
    RMD_LAUNCH_CHECK("atrous_stream_kernel");
    return RMD_OK;
}

template <int S, int NP>
static int launch_planned(const AtrousArgs& a, hipStream_t stream)
{
    // the byte-storing form exists for the library's default decomposition only (NP = 2); launch_atrous routes every
    // other form through a float plane and a conversion launch
    if constexpr (NP == 2) { if (a.out8) return launch_planned_as<S, NP, true>(a, stream); }
    return launch_planned_as<S, NP, false>(a, stream);
}

template <int S, int NP>
static int launch_stream(AtrousArgs a, hipStream_t stream)
{
    const int b0 = a.b_row0, b1 = a.b_row1;
    a.b_nblocks = 0;
    if (b1 > b0) {              // second range: planned on its own, appended to the first in the workgroup order
        AtrousArgs b = a;
        b.row0 = b0; b.row1 = b1;
        plan_stream<S, NP>(b);
        a.b_band_h = b.band_h; a.b_band_base = b.band_base; a.b_nblocks = b.nblocks;
        a.b_n_hi = b.n_hi; a.b_band_h_hi = b.band_h_hi; a.b_nblocks_hi = b.nblocks_hi;
    }
    plan_stream<S, NP>(a);
    a.per_xcd = (a.nblocks + a.b_nblocks + kXcds - 1) / kXcds;
    return launch_planned<S, NP>(a, stream);
}

// Library default: 128-column strips with two row pairs per step (NP = 2).  The other two decompositions were
// measured against it with bench.py's timing and never won: four row pairs on 64-column strips (NP = 4: 127 /
// 132 / 131 / 139 / 142 us per 4K iteration against 125 / 131 / 132 / 133 / 140) and one row pair on 256-column
// strips (NP = 1, two workgroups per CU: 84 us per iteration on a 7680x540 strip against 74-80, and 84.5 against
// 86.4 only at step 16, where NP = 2 has 960 workgroups for 768 slots).  atrous_variant 2 / 6 of the experiments build run them.
template <int S>
static int launch_stream_auto(AtrousArgs a, hipStream_t stream)
{
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
    a.b_row0 = a.b_row1 = a.b_band_h = a.b_band_base = a.b_nblocks = a.b_n_hi = a.b_band_h_hi = a.b_nblocks_hi = 0;
    a.albedo8 = nullptr; a.out8 = nullptr;
#ifdef RMD_EXPERIMENTS
    a.side_units = 0;
#endif
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
#ifdef RMD_EXPERIMENTS
    RMD_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_atrous_lc_errors), sizeof(unsigned int)));
#else
    *count = 0;     // variant 7 (the only kernel with a counter protocol) is not in this build
#endif
    return RMD_OK;
}

extern "C" int rmd_svgf_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                               const float* in, float* out, int row0, int row1, void* stream)
{
    return rmd_svgf_atrous2(f, p, iteration, in, out, row0, row1, 0, 0, stream);
}

extern "C" int rmd_svgf_atrous2(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                                const float* in, float* out, int row0, int row1, int row0b, int row1b, void* stream)
{
    return launch_atrous(f, p, iteration, in, out, row0, row1, row0b, row1b, stream, nullptr, nullptr);
}

int rmd::launch_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration, const float* in, float* out,
                       int row0, int row1, int row0b, int row1b, void* stream, const AtrousSide* side, const GBuffer8* g8)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_atrous: params is NULL");
    // g8: the iteration stores modulated, quantised bytes to g8->denoised instead of floats to `out` (which may be NULL)
    if (g8) {
        if (!g8->albedo || !g8->denoised) return fail(RMD_E_NULL, "rmd_svgf_gbuffer_frame: albedo / denoised plane is NULL");
        if (!aligned_to(g8->albedo, 4) || !aligned_to(g8->denoised, 4)) return fail(RMD_E_ALIGN, "rmd_svgf_gbuffer_frame: uchar4 planes must be 4-byte aligned");
        if (side || row1b > row0b || !(p->atrous_variant == 0 || p->atrous_variant == 1 || p->atrous_variant == 3))
            return fail(RMD_E_UNSUPPORTED, "rmd_svgf_gbuffer_frame: only the default / direct a-trous kernels store bytes");
        if (!out) out = reinterpret_cast<float*>(g8->denoised);       // (not written; keeps the checks below meaningful)
    }
    if (!in || !out || !f->nd) return fail(RMD_E_NULL, "rmd_svgf_atrous: in/out/nd plane is NULL");
    if (in == out) return fail(RMD_E_BUFFER, "rmd_svgf_atrous: in and out alias (taps cross pixels)");
    if (iteration < 0 || iteration > 12) return fail(RMD_E_PARAM, "rmd_svgf_atrous: iteration %d outside [0,12]", iteration);
    if (!(p->sigma_n > 0.0f) || !(p->sigma_z > 0.0f) || !(p->sigma_l > 0.0f))
        return fail(RMD_E_PARAM, "rmd_svgf_atrous: sigma_n/sigma_z/sigma_l must be > 0");
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_atrous: rows [%d,%d) invalid", row0, row1);
    const int s = 1 << iteration;
    if (int e = check_rows_in_buffer(f, row0 - 2 * s, row1 + 2 * s, "rmd_svgf_atrous")) return e;
    const bool two = row1b > row0b;
    // only the default row-streaming kernel takes two row ranges in one launch; every other form runs them one after the other
    const bool streams = (p->atrous_variant == 0 || p->atrous_variant == 3) && iteration <= 4;
    if (side && !streams) return fail(RMD_E_UNSUPPORTED, "rmd_svgf_frame_atrous_next: only the row-streaming a-trous kernel (atrous_variant 0, iterations 0..4) carries a side job");
    if (two && !streams) {
        if (int e = rmd_svgf_atrous(f, p, iteration, in, out, row0, row1, stream)) return e;
        return rmd_svgf_atrous(f, p, iteration, in, out, row0b, row1b, stream);
    }
    if (two) {
        if (row0b < row1 || row1b > f->height) return fail(RMD_E_ROWS, "rmd_svgf_atrous2: second range [%d,%d) must lie behind the first [%d,%d)", row0b, row1b, row0, row1);
        if (int e = check_rows_in_buffer(f, row0b - 2 * s, row1b + 2 * s, "rmd_svgf_atrous2")) return e;
    }
    if (!aligned_to(in, 16) || (!g8 && !aligned_to(out, 16)) || !aligned_to(f->nd, 16))
        return fail(RMD_E_ALIGN, "rmd_svgf_atrous: float4 planes must be 16-byte aligned");

    AtrousArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.in = (const float4*)in; a.nd = (const float4*)f->nd; a.out = (float4*)out;
    a.row0 = row0; a.row1 = row1; a.step = s;
    a.sigma_n = p->sigma_n; a.sigma_z = p->sigma_z; a.sigma_l = p->sigma_l;
    a.band_h = a.band_base = a.nstrips = a.nblocks = a.per_xcd = 0;
    a.n_int = a.xe_lo = a.band_h_xe = a.total_int = a.int_per_xcd = a.xe_per_xcd = 0;
    a.n_hi = a.band_h_hi = a.nblocks_hi = 0;
    a.b_row0 = two ? row0b : 0; a.b_row1 = two ? row1b : 0;
    a.b_band_h = a.b_band_base = a.b_nblocks = a.b_n_hi = a.b_band_h_hi = a.b_nblocks_hi = 0;
    a.nt_out = (double)(row1 - row0 + (two ? row1b - row0b : 0)) * f->width * 48.0 > 256.0e6 ? 1 : 0;
    a.nt_out = tuning_env("RMD_NT_OUT", a.nt_out);          // (A/B knob of tools/boundary_probe.py; experiments build only)
    a.albedo8 = g8 ? g8->albedo : nullptr; a.out8 = g8 ? g8->denoised : nullptr;
#ifdef RMD_EXPERIMENTS
    a.side_units = 0; a.side_every = 1; a.side_counter = nullptr;
    if (side && side->units > 0) { a.side = side->t; a.side_counter = side->counter; a.side_units = side->units; a.side_every = side->every < 1 ? 1 : side->every; }
#else
    if (side) return fail(RMD_E_UNSUPPORTED, "rmd_svgf_frame_atrous_next is an experiment (make experiments; rmd_has_experiments() == 0 here)");
#endif
    if (p->atrous_cus < 0) return fail(RMD_E_PARAM, "rmd_svgf_atrous: atrous_cus %d is negative", p->atrous_cus);
    a.cus = p->atrous_cus > 0 && p->atrous_cus < device_cus() ? p->atrous_cus : device_cus();

    int variant = p->atrous_variant;
    if (variant == 0) {
        if (iteration <= 4) return launch_stream_auto_iter(iteration, a, as_stream(stream));
        variant = 1;
    }
    if (variant == 3) {                                                               // 128 columns x 2 row pairs (= the default for iterations 0..4)
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        return launch_stream_iter<2>(iteration, a, as_stream(stream));
    }
    if (variant == 2 || (variant >= 4 && variant <= 8)) {
#ifdef RMD_EXPERIMENTS
        // formulations that were measured and lost (DESIGN.md sections 4.4-4.7); experiments build only
        if (iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variants cover iterations 0..4");
        switch (variant) {
            case 2: return launch_stream_iter<1>(iteration, a, as_stream(stream));     // 256 columns x 1 row pair
            case 6: return launch_stream_iter<4>(iteration, a, as_stream(stream));     // 64 columns x 4 row pairs
            case 4: return launch_pair_iter(iteration, a, as_stream(stream));          // pixel pairs, SoA LDS planes
            case 7: return launch_lc_iter(iteration, a, as_stream(stream));            // pixel pairs, loader waves + compute waves
            case 8: return launch_quad_iter(iteration, a, as_stream(stream));          // 2 x 2 pixel block per lane
            default: {
                dim3 grid5((f->width + 63) / 64, (row1 - row0 + 3) / 4);
                hipLaunchKernelGGL(atrous_direct2_kernel, grid5, dim3(256), 0, as_stream(stream), a);
                RMD_LAUNCH_CHECK("atrous_direct2_kernel");
                return RMD_OK;
            }
        }
#else
        return fail(RMD_E_UNSUPPORTED, "rmd_svgf_atrous: atrous_variant %d is an experiment (make experiments; rmd_has_experiments() == 0 here)", variant);
#endif
    }
    if (variant != 1) return fail(RMD_E_PARAM, "rmd_svgf_atrous: unknown atrous_variant %d", p->atrous_variant);
    dim3 grid((f->width + 63) / 64, (row1 - row0 + 3) / 4);
    hipLaunchKernelGGL(atrous_direct_kernel, grid, dim3(256), 0, as_stream(stream), a);
    RMD_LAUNCH_CHECK("atrous_direct_kernel");
    return RMD_OK;
}
