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
//            workgroup (4 wave64 = 256 threads) owns a 256-pixel-wide column strip of one
//            lattice inside one band of rows and walks down it; a ring of 6 lattice rows
//            (color + nd, width 256 + 4S) lives in LDS, so every input row is fetched from
//            L2/HBM once per strip as 16-byte-per-lane coalesced segments and each of the 25
//            taps is a conflict-free ds_read_b128 with an immediate offset.  Each thread
//            produces two vertically adjacent lattice pixels per step (30 tap fetches for 50
//            weight evaluations), the next two lattice rows are prefetched into registers while
//            the current pair is computed.  Workgroup ids are remapped so that each XCD owns a
//            contiguous run of (band, strip, lattice) work: neighbouring strips / lattices, which
//            share halo columns and the +-1 variance rows, hit the same 4 MiB L2.
//            Algorithmic traffic 48 B/px/iteration (32 read + 16 written), SURVEY §8d.
//
// The weights are evaluated in the log2 domain (one v_log_f32 + one v_exp_f32 per tap):
//   w = exp2( log2 k + sigma_n*log2(max(0,n_p.n_t)) - |dz|*log2e/(za*len+1e-8) - |dl|*log2e/l_den )
#include "common.h"
#include <type_traits>

#ifndef RMD_ATROUS_COLBUF
#define RMD_ATROUS_COLBUF 1   // tap columns in flight: 2 = fetch next while weighting current (spills at 256 VGPRs)
#endif

namespace rmd {

struct AtrousArgs {
    Geom g;
    const float4* in; const float4* nd; float4* out;
    int row0, row1;
    int step;
    float sigma_n, sigma_z, sigma_l;
    // stream variant work decomposition
    int band_h, nstrips, nblocks, per_xcd;
};

// log2 of the B3-spline taps {3/8, 1/4, 1/16} (reference src/filter.cu:10)
__device__ constexpr float kLogB3[3] = { -1.41503749927884381855f, -2.0f, -4.0f };
// 3x3 Gaussian variance prefilter: centre, edge, corner
__device__ constexpr float kG3[3] = { 0.25f, 0.125f, 0.0625f };

struct Center {
    float nx, ny, nz, z, lum, il;
    float iz[5];      // log2e/(za*len+1e-8) for len = 1, sqrt2, 2, sqrt5, 2*sqrt2
    bool zero;        // centre normal is (0,0,0)
};
struct Acc { float sw, sr, sg, sb, sv; };

__device__ __forceinline__ int len_class(int adx, int ady)
{
    // (adx,ady) in {0,1,2}^2 minus (0,0) -> index into Center::iz
    const int m = adx * adx + ady * ady;            // 1,2,4,5,8
    return m == 1 ? 0 : m == 2 ? 1 : m == 4 ? 2 : m == 5 ? 3 : 4;
}

__device__ __forceinline__ Center make_center(const float4 c, const float4 n, float var_c, float gz,
                                              float sigma_z, float sigma_l, float step)
{
    Center k;
    k.nx = n.x; k.ny = n.y; k.nz = n.z; k.z = n.w;
    k.zero = is_zero3(n);
    k.lum = lum3(c.x, c.y, c.z);
    const float vc = var_c > 0.0f ? var_c : 0.0f;
    k.il = kLog2e * fast_rcp(sigma_l * sqrtf(vc) + 1e-8f);
    const float za = sigma_z * fmaxf(gz, 1e-8f) * step;
    k.iz[0] = kLog2e * fast_rcp(za * 1.0f + 1e-8f);
    k.iz[1] = kLog2e * fast_rcp(za * 1.41421356237309504880f + 1e-8f);
    k.iz[2] = kLog2e * fast_rcp(za * 2.0f + 1e-8f);
    k.iz[3] = kLog2e * fast_rcp(za * 2.23606797749978969641f + 1e-8f);
    k.iz[4] = kLog2e * fast_rcp(za * 2.82842712474619009760f + 1e-8f);
    return k;
}

// One tap.  e0 = log2 k (or -inf for a tap outside the frame: w becomes exactly 0).
// ZERO_AWARE=false assumes the centre normal is non-zero (a zero tap normal then gives
// log2(0) = -inf -> w = 0, as Appendix A.A.2 requires); true also handles zero centres.
template <bool ZERO_AWARE>
__device__ __forceinline__ void tap_accum(Acc& s, const Center& k, const float4 tc, const float4 tn, float tl,
                                          float e0, int adx, int ady, float sigma_n)
{
    const float d = __builtin_fmaf(k.nz, tn.z, __builtin_fmaf(k.ny, tn.y, k.nx * tn.x));
    float e = __builtin_fmaf(sigma_n, fast_log2(fmaxf(d, 0.0f)), e0);
    if (ZERO_AWARE) {
        const bool tz = is_zero3(tn);
        if (k.zero) e = tz ? e0 : kNegInf;
    }
    if (adx | ady) e = __builtin_fmaf(-fabsf(k.z - tn.w), k.iz[len_class(adx, ady)], e);
    e = __builtin_fmaf(-fabsf(k.lum - tl), k.il, e);
    const float w = fast_exp2(e);
    s.sw += w;
    s.sr = __builtin_fmaf(w, tc.x, s.sr);
    s.sg = __builtin_fmaf(w, tc.y, s.sg);
    s.sb = __builtin_fmaf(w, tc.z, s.sb);
    s.sv = __builtin_fmaf(w * w, tc.w, s.sv);
}

__device__ __forceinline__ float4 finish(const Acc& s, const float4 c)
{
    if (s.sw < 1e-10f) return c;                       // A.A.3 pass-through
    const float inv = fast_rcp(s.sw);
    return make_float4(s.sr * inv, s.sg * inv, s.sb * inv, s.sv * inv * inv);
}

// ---------------------------------------------------------------------------------- direct
__global__ __launch_bounds__(256) void atrous_direct_kernel(AtrousArgs a)
{
    const Geom g = a.g;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= a.row1) return;
    const int s = a.step;
    const size_t i = pix_index(g, x, y);
    const float4 c = a.in[i];
    const float4 n = a.nd[i];

    // A.A.1: 3x3 Gaussian prefilter of the variance, OOB skipped + renormalised
    float gsum = 0.0f, vsum = 0.0f;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int tx = x + dx, ty = y + dy;
            if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) continue;
            const float gk = kG3[(dx != 0) + (dy != 0)];
            gsum += gk;
            vsum += gk * a.in[pix_index(g, tx, ty)].w;
        }
    const float var_c = vsum / gsum;
    const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
    const float gz = fabsf(a.nd[pix_index(g, x1, y)].w - n.w) + fabsf(a.nd[pix_index(g, x, y1)].w - n.w);
    const Center k = make_center(c, n, var_c, gz, a.sigma_z, a.sigma_l, (float)s);

    Acc acc = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int tx = x + s * dx, ty = y + s * dy;
            if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) continue;
            const size_t ti = pix_index(g, tx, ty);
            const float4 tc = a.in[ti];
            const float4 tn = a.nd[ti];
            const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            const float e0 = kLogB3[adx] + kLogB3[ady];
            tap_accum<true>(acc, k, tc, tn, lum3(tc.x, tc.y, tc.z), e0, adx, ady, a.sigma_n);
        }
    }
    a.out[i] = finish(acc, c);
}

// ---------------------------------------------------------------------------------- stream
template <int S>
struct StreamCfg {
    static constexpr int BX = 256;                    // output columns per workgroup = threads
    static constexpr int PW = BX + 4 * S;             // staged row width in pixels (halo 2S each side)
    static constexpr int NR = 6;                      // ring: lattice rows j-2 .. j+3
    static constexpr int ROW_BYTES = PW * 16;
    static constexpr int PLANE_BYTES = NR * ROW_BYTES;
    static constexpr int VAR_OFF = 2 * PLANE_BYTES;   // 4 rows of BX+2 floats: variance of rows y-1 / y+1
    static constexpr int VAR_ROW = BX + 2;
    static constexpr int LDS_BYTES = VAR_OFF + 4 * VAR_ROW * 4;
};

__device__ __forceinline__ float4 lds_f4(const unsigned char* lds, int off) { return *reinterpret_cast<const float4*>(lds + off); }
__device__ __forceinline__ float  lds_f1(const unsigned char* lds, int off) { return *reinterpret_cast<const float*>(lds + off); }

template <int S, bool EDGE>
__device__ __forceinline__ void atrous_stream_body(const AtrousArgs& a, unsigned char* lds, const int tid,
                                                   const int x0, const int ybase, const int nj)
{
    using C = StreamCfg<S>;
    const Geom g = a.g;
    const int x = x0 + tid;
    const bool xin = !EDGE || x < g.W;
    const float* in_f = reinterpret_cast<const float*>(a.in);
    const float* nd_f = reinterpret_cast<const float*>(a.nd);
    float* var_lds = reinterpret_cast<float*>(lds + C::VAR_OFF);

    // ---- register prefetch state: two lattice rows (color, nd), and for the next output pair the
    // variance of rows y-1 / y+1 (3x3 prefilter) and z of row y+1 (depth gradient)
    float4 pc[2], pn[2], pe;
    float pvu[2], pvd[2], pzd[2], pvh = 0.0f;
    float zd_cur[2] = { 0.0f, 0.0f };

    // Global addresses are formed as (wave-uniform row base) + (lane index): the uniform part
    // stays in SGPRs (global_load ... saddr), only tid*16 lives in a VGPR.
    auto row_base = [&](const int y, const int xs) -> long long {
        return (long long)(y - g.buf_row0) * (long long)g.W + (long long)xs;
    };
    // A staged row is BX + 4S pixels.  Threads load pixel `tid` of both rows and both planes; the
    // 4S-pixel tails of the 2 rows x 2 planes (16S float4 in all) are spread over lanes < 16S as
    // ONE extra load per lane: tail element e -> row (e/4S)>>1, plane (e/4S)&1, column BX + e%4S.
    const int e_sel = tid / (4 * S), e_col = C::BX + tid % (4 * S);
    const bool e_act = tid < 16 * S;
    auto load_rows = [&](const int j0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = j0 + i;
            const int y = ybase + j * S;
            const bool rowok = (j <= nj + 1) && (!EDGE || (y >= 0 && y < g.H));
            bool act = rowok;
            if (EDGE) act = act && x - 2 * S >= 0 && x - 2 * S < g.W;
            float4 vc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vn = vc;
            if (act) {
                const long long o = row_base(y, x0 - 2 * S);
                vc = (a.in + o)[tid];
                vn = (a.nd + o)[tid];
            }
            pc[i] = vc;
            pn[i] = vn;
        }
        {
            const int j = j0 + (e_sel >> 1);
            const int y = ybase + j * S;
            const int gx = x0 - 2 * S + e_col;
            bool act = e_act && (j <= nj + 1);
            if (EDGE) act = act && y >= 0 && y < g.H && gx >= 0 && gx < g.W;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (act) {
                const float4* plane = (e_sel & 1) ? a.nd : a.in;
                v = plane[(size_t)(y - g.buf_row0) * (size_t)g.W + (size_t)gx];
            }
            pe = v;
        }
    };
    auto store_rows = [&](const int j0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = (j0 + i + 6) % 6;
            const int off = slot * C::ROW_BYTES + tid * 16;
            *reinterpret_cast<float4*>(lds + off) = pc[i];
            *reinterpret_cast<float4*>(lds + C::PLANE_BYTES + off) = pn[i];
        }
        if (e_act) {
            const int slot = (j0 + (e_sel >> 1) + 6) % 6;
            const int off = ((e_sel & 1) ? C::PLANE_BYTES : 0) + slot * C::ROW_BYTES + e_col * 16;
            *reinterpret_cast<float4*>(lds + off) = pe;
        }
    };
    auto load_aux = [&](const int jo) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = ybase + (jo + i) * S;
            const bool rowok = (jo + i) < nj && xin;
            float vu = 0.0f, vd = 0.0f, zd = 0.0f;
            if (rowok) {
                if (!EDGE || y - 1 >= 0) vu = (in_f + row_base(y - 1, x0) * 4 + 3)[tid * 4];
                if (!EDGE || y + 1 < g.H) vd = (in_f + row_base(y + 1, x0) * 4 + 3)[tid * 4];
                const int yz = EDGE ? min(y + 1, g.H - 1) : y + 1;
                zd = (nd_f + row_base(yz, x0) * 4 + 3)[tid * 4];
            }
            pvu[i] = vu; pvd[i] = vd; pzd[i] = zd;
        }
        pvh = 0.0f;
        if (tid < 8) {   // halo columns x0-1 and x0+BX of the four variance rows
            const int ii = tid >> 2, ud = (tid >> 1) & 1, side = tid & 1;
            const int yy = ybase + (jo + ii) * S + (ud ? 1 : -1);
            const int xx = side ? x0 + C::BX : x0 - 1;
            bool ok = (jo + ii) < nj;
            if (EDGE) ok = ok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            if (ok) pvh = in_f[((size_t)(yy - g.buf_row0) * (size_t)g.W + (size_t)xx) * 4 + 3];
        }
    };
    auto store_aux = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            var_lds[(i * 2 + 0) * C::VAR_ROW + tid + 1] = pvu[i];
            var_lds[(i * 2 + 1) * C::VAR_ROW + tid + 1] = pvd[i];
            zd_cur[i] = pzd[i];
        }
        if (tid < 8) {
            const int ii = tid >> 2, ud = (tid >> 1) & 1, side = tid & 1;
            var_lds[(ii * 2 + ud) * C::VAR_ROW + (side ? C::BX + 1 : 0)] = pvh;
        }
    };

    // ---- per-pixel setup (A.A.1 prefilter, depth gradient) from the staged data
    auto setup = [&](const int i, const int y, const int rb_center, const float4 c, const float4 n) -> Center {
        const int ccol = rb_center + 2 * S * 16;      // byte offset of the centre pixel in the color plane
        const float* vu = var_lds + (i * 2 + 0) * C::VAR_ROW + tid;     // [0..2] = x-1, x, x+1 of row y-1
        const float* vd = var_lds + (i * 2 + 1) * C::VAR_ROW + tid;
        const float v_l = lds_f1(lds, ccol - 16 + 12), v_r = lds_f1(lds, ccol + 16 + 12);
        float var_c;
        if (!EDGE) {
            // oracle order: dx outer, dy inner; weights sum to exactly 1
            float vs = kG3[2] * vu[0];
            vs += kG3[1] * v_l;  vs += kG3[2] * vd[0];
            vs += kG3[1] * vu[1]; vs += kG3[0] * c.w; vs += kG3[1] * vd[1];
            vs += kG3[2] * vu[2]; vs += kG3[1] * v_r; vs += kG3[2] * vd[2];
            var_c = vs;
        } else {
            const bool okl = x - 1 >= 0, okr = x + 1 < g.W, oku = y - 1 >= 0, okd = y + 1 < g.H;
            float gs = 0.0f, vs = 0.0f;
            if (okl && oku) { gs += kG3[2]; vs += kG3[2] * vu[0]; }
            if (okl)        { gs += kG3[1]; vs += kG3[1] * v_l; }
            if (okl && okd) { gs += kG3[2]; vs += kG3[2] * vd[0]; }
            if (oku)        { gs += kG3[1]; vs += kG3[1] * vu[1]; }
            gs += kG3[0]; vs += kG3[0] * c.w;
            if (okd)        { gs += kG3[1]; vs += kG3[1] * vd[1]; }
            if (okr && oku) { gs += kG3[2]; vs += kG3[2] * vu[2]; }
            if (okr)        { gs += kG3[1]; vs += kG3[1] * v_r; }
            if (okr && okd) { gs += kG3[2]; vs += kG3[2] * vd[2]; }
            var_c = vs / gs;
        }
        float zr = lds_f1(lds, C::PLANE_BYTES + ccol + 16 + 12);
        if (EDGE && !(x + 1 < g.W)) zr = n.w;
        const float gz = fabsf(zr - n.w) + fabsf(zd_cur[i] - n.w);
        return make_center(c, n, var_c, gz, a.sigma_z, a.sigma_l, (float)S);
    };

    // ---- one step: outputs at lattice rows j and j+1
    auto compute = [&](const int j) {
        int rb[6];
        {
            int slot = (j + 4) % 6;                    // ring slot of lattice row j-2
#pragma unroll
            for (int tr = 0; tr < 6; ++tr) {
                rb[tr] = slot * C::ROW_BYTES + tid * 16;
                slot = slot == 5 ? 0 : slot + 1;
            }
        }
        const int yA = ybase + j * S, yB = yA + S;
        const float4 cA = lds_f4(lds, rb[2] + 2 * S * 16), nA = lds_f4(lds, C::PLANE_BYTES + rb[2] + 2 * S * 16);
        const float4 cB = lds_f4(lds, rb[3] + 2 * S * 16), nB = lds_f4(lds, C::PLANE_BYTES + rb[3] + 2 * S * 16);
        const Center kA = setup(0, yA, rb[2], cA, nA);
        const Center kB = setup(1, yB, rb[3], cB, nB);
        Acc sA = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, sB = sA;

        bool rowv[6];
#pragma unroll
        for (int tr = 0; tr < 6; ++tr) {
            const int yy = yA + (tr - 2) * S;
            rowv[tr] = !EDGE || (yy >= 0 && yy < g.H);
        }
        // a wave takes the cheaper path when none of its centres has a zero normal
        const bool any_zero = __builtin_amdgcn_ballot_w64(kA.zero || kB.zero) != 0ull;

        // One tap column (fixed dx, the six ring rows) is fetched while the previous column is
        // being weighted; sched_barrier keeps the compiler from hoisting all 60 ds_read_b128 of a
        // step to the top (which needs > 256 VGPRs and spills).
        auto load_col = [&](const int dxi, float4 (&tc)[6], float4 (&tn)[6]) {
#pragma unroll
            for (int tr = 0; tr < 6; ++tr) {
                const int off = rb[tr] + (2 * S + (dxi - 2) * S) * 16;
                tc[tr] = lds_f4(lds, off);
                tn[tr] = lds_f4(lds, C::PLANE_BYTES + off);
            }
        };
        auto taps = [&](auto zero_aware) {
            constexpr bool ZA = decltype(zero_aware)::value;
            auto weigh_col = [&](const int dxi, const float4 (&tc)[6], const float4 (&tn)[6]) {
                const int dx = dxi - 2;
                const int adx = dx < 0 ? -dx : dx;
                const bool colv = !EDGE || (x + dx * S >= 0 && x + dx * S < g.W);
#pragma unroll
                for (int tr = 0; tr < 6; ++tr) {
                    const float tl = lum3(tc[tr].x, tc[tr].y, tc[tr].z);
                    const float ecol = (!EDGE || (colv && rowv[tr])) ? kLogB3[adx] : kNegInf;
                    if (tr <= 4) {
                        const int dy = tr - 2, ady = dy < 0 ? -dy : dy;
                        tap_accum<ZA>(sA, kA, tc[tr], tn[tr], tl, ecol + kLogB3[ady], adx, ady, a.sigma_n);
                    }
                    if (tr >= 1) {
                        const int dy = tr - 3, ady = dy < 0 ? -dy : dy;
                        tap_accum<ZA>(sB, kB, tc[tr], tn[tr], tl, ecol + kLogB3[ady], adx, ady, a.sigma_n);
                    }
                }
            };
#if RMD_ATROUS_COLBUF == 2
            float4 c0[6], n0[6], c1[6], n1[6];
            load_col(0, c0, n0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dxi = 0; dxi < 5; ++dxi) {
                if (dxi & 1) { if (dxi < 4) load_col(dxi + 1, c0, n0); weigh_col(dxi, c1, n1); }
                else         { if (dxi < 4) load_col(dxi + 1, c1, n1); weigh_col(dxi, c0, n0); }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int dxi = 0; dxi < 5; ++dxi) {
                float4 c0[6], n0[6];
                load_col(dxi, c0, n0);
                weigh_col(dxi, c0, n0);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        };
        if (any_zero) taps(std::true_type{}); else taps(std::false_type{});

        if (xin) {
            (a.out + row_base(yA, x0))[tid] = finish(sA, cA);
            if (j + 1 < nj) (a.out + row_base(yB, x0))[tid] = finish(sB, cB);
        }
    };

    // ---- prologue: lattice rows -2..3 and the aux rows of the first pair
    load_rows(-2); store_rows(-2);
    load_rows(0);  store_rows(0);
    load_rows(2);  load_aux(0);
    store_rows(2); store_aux();
    __syncthreads();

    for (int j = 0; j < nj; j += 2) {
        const bool more = j + 2 < nj;
        if (more) { load_rows(j + 4); load_aux(j + 2); }     // in flight during compute
        compute(j);
        if (!more) break;
        __syncthreads();                                     // every wave is done reading rows j-2, j-1
        store_rows(j + 4); store_aux();
        __syncthreads();
    }
}

template <int S>
__global__ __launch_bounds__(256, 2) void atrous_stream_kernel(AtrousArgs a)
{
    using C = StreamCfg<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char atrous_lds[];
    const int tid = threadIdx.x;
    // XCD-aware remap: workgroups pid, pid+8, ... share an XCD (round-robin dispatch), give each
    // XCD one contiguous run of logical work so halo columns / variance rows are shared in its L2.
    const int pid = blockIdx.x;
    const int L = (pid & (kXcds - 1)) * a.per_xcd + (pid >> 3);
    if (L >= a.nblocks) return;
    const int r = L % S, t = L / S;
    const int strip = t % a.nstrips, band = t / a.nstrips;
    const int x0 = strip * C::BX;
    const int yb = a.row0 + band * a.band_h;
    const int ye = min(yb + a.band_h, a.row1);
    const int ybase = yb + r;
    if (ybase >= ye) return;
    const int nj = (ye - ybase + S - 1) / S;
    const bool edge = (x0 - 2 * S < 0) || (x0 + C::BX + 2 * S > a.g.W) || (ybase - 2 * S < 0) ||
                      (ybase + (nj + 1) * S >= a.g.H);
    if (edge) atrous_stream_body<S, true>(a, atrous_lds, tid, x0, ybase, nj);
    else      atrous_stream_body<S, false>(a, atrous_lds, tid, x0, ybase, nj);
}

template <int S>
static int launch_stream(AtrousArgs a, hipStream_t stream)
{
    using C = StreamCfg<S>;
    static bool attr_done = false;
    if (!attr_done) {
        RMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&atrous_stream_kernel<S>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_done = true;
    }
    const int rows = a.row1 - a.row0;
    a.nstrips = (a.g.W + C::BX - 1) / C::BX;
    // one resident wave of workgroups: 2 per CU (LDS-limited) x 256 CUs
    const int per_band = a.nstrips * S;
    int nb = (2 * kCus) / per_band;
    if (nb < 1) nb = 1;
    int bh = (rows + nb - 1) / nb;
    bh = ((bh + 2 * S - 1) / (2 * S)) * (2 * S);       // whole output pairs per lattice
    a.band_h = bh;
    const int nbands = (rows + bh - 1) / bh;
    a.nblocks = nbands * per_band;
    a.per_xcd = (a.nblocks + kXcds - 1) / kXcds;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(atrous_stream_kernel<S>), dim3(a.per_xcd * kXcds), dim3(C::BX), C::LDS_BYTES, stream, a);
    RMD_LAUNCH_CHECK("atrous_stream_kernel");
    return RMD_OK;
}

}  // namespace rmd

using namespace rmd;

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
    a.band_h = a.nstrips = a.nblocks = a.per_xcd = 0;

    int variant = p->atrous_variant;
    if (variant == 0) variant = (iteration <= 4) ? 2 : 1;
    if (variant == 2 && iteration > 4) return fail(RMD_E_PARAM, "rmd_svgf_atrous: the stream variant covers iterations 0..4");
    if (variant == 2) {
        switch (iteration) {
            case 0: return launch_stream<1>(a, as_stream(stream));
            case 1: return launch_stream<2>(a, as_stream(stream));
            case 2: return launch_stream<4>(a, as_stream(stream));
            case 3: return launch_stream<8>(a, as_stream(stream));
            default: return launch_stream<16>(a, as_stream(stream));
        }
    }
    if (variant != 1) return fail(RMD_E_PARAM, "rmd_svgf_atrous: unknown atrous_variant %d", p->atrous_variant);
    dim3 grid((f->width + 63) / 64, (row1 - row0 + 3) / 4);
    hipLaunchKernelGGL(atrous_direct_kernel, grid, dim3(256), 0, as_stream(stream), a);
    RMD_LAUNCH_CHECK("atrous_direct_kernel");
    return RMD_OK;
}
