// svgf_variance.hip — V pass: spatial variance estimate for pixels with a short history
// (SURVEY Appendix A.V).  Not in the reference (README.md:3-10 names "variance-guided
// filtering"; no code exists).
//
// One thread per pixel, a wave owns 64 consecutive x.  Pixels with h >= var_h_threshold copy
// T's output (48 B read + 16 B written per pixel = the 64 B/px of SURVEY §8d).  Pixels on the
// spatial path run a (2R+1)^2 edge-stopped window from global memory: those pixels are spatially
// coherent (disocclusion bands, the first frames), so whole waves take one branch and the window
// re-reads are served by L1/L2.  Frame statistics (sum of variance, spatial-path pixel count,
// sum of history length, pixel count) are reduced across the wavefront with __shfl_xor
// butterflies and committed with one atomic per wave.
#include <cstdlib>
#include "common.h"
#include "svgf_tv.h"

namespace rmd {

struct VarianceArgs {
    Geom g;
    const float4* t_color; const unsigned char* t_len; const float4* nd;
    float4* v_color; float* stats;
    int row0, row1;
    int h_threshold, radius;
    int prefilled;            // v_color already holds t_color (fused frame): long-history pixels are left alone
    const unsigned char* tile_flags;   // fused frame: tiles T marked as holding short-history pixels
    int tiles_x;
    int sparse_t_color;       // T wrote t_color only inside flagged tiles: elsewhere v_color holds the same values
    float sigma_n, sigma_z;
};

// RFIX > 0: the window radius as a compile-time constant, so the dy loop unrolls and the 2*(2R+1)
// gathers of a window column are in flight together (the few waves that run the window in the
// steady state are latency-bound).  RFIX = 0 keeps the generic loop for other radii.
//
// One pixel of tile (tile_x, tile_y) of the GLOBAL 64x4 tiling, the same tiles T flags.  Returns the
// pixel's contribution to the frame statistics in s[4] (zeros for a pixel it does not visit).
template <int RFIX>
__device__ __forceinline__ void variance_pixel(const VarianceArgs& a, const int tile_x, const int tile_y, float (&s)[4])
{
    const Geom g = a.g;
    const int x = tile_x * 64 + (threadIdx.x & 63);
    const int y = tile_y * 4 + (threadIdx.x >> 6);
    const bool active = x < g.W && y >= a.row0 && y < a.row1;

    float s_var = 0.0f, s_spatial = 0.0f, s_h = 0.0f, s_n = 0.0f;
    s[0] = s[1] = s[2] = s[3] = 0.0f;
    if (active) {
        const size_t i = pix_index(g, x, y);
        const int h = (int)a.t_len[i];
        const bool spatial = h < a.h_threshold;
        if (!spatial && a.prefilled) return;       // (prefilled launches collect no statistics; s stays 0)
        const float4 c = a.t_color[i];
        float4 o = c;
        if (spatial) {
            const float4 nd = a.nd[i];
            const int x1 = min(x + 1, g.W - 1), y1 = min(y + 1, g.H - 1);
            const float gz = fabsf(a.nd[pix_index(g, x1, y)].w - nd.w) + fabsf(a.nd[pix_index(g, x, y1)].w - nd.w);
            const float za = a.sigma_z * fmaxf(gz, 1e-8f);
            const bool p_zero = is_zero3(nd);
            float sw = 0.0f, scx = 0.0f, scy = 0.0f, scz = 0.0f, sl = 0.0f, sl2 = 0.0f;
            auto tap = [&](const int dx, const int dy) {
                const int tx = x + dx, ty = y + dy;
                if (tx < 0 || tx >= g.W || ty < 0 || ty >= g.H) return;
                const size_t ti = pix_index(g, tx, ty);
                const float4 tc = a.t_color[ti];
                const float4 tn = a.nd[ti];
                // log2-domain edge-stopping weight: w = exp2(sigma_n*log2(max(0,n.n)) - w_z*log2 e)
                float e;
                const bool t_zero = is_zero3(tn);
                if (p_zero || t_zero) {
                    e = (p_zero && t_zero) ? 0.0f : kNegInf;
                } else {
                    const float d = __builtin_fmaf(nd.z, tn.z, __builtin_fmaf(nd.y, tn.y, nd.x * tn.x));
                    e = a.sigma_n * fast_log2(fmaxf(d, 0.0f));
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
            };
            // tap order dx outer / dy inner in both forms
            if constexpr (RFIX > 0) {
                for (int dx = -RFIX; dx <= RFIX; ++dx) {
#pragma unroll
                    for (int dy = -RFIX; dy <= RFIX; ++dy) tap(dx, dy);
                }
            } else {
                const int R = a.radius;
                for (int dx = -R; dx <= R; ++dx)
                    for (int dy = -R; dy <= R; ++dy) tap(dx, dy);
            }
            if (!(sw < 1e-10f)) {
                const float el = sl / sw, el2 = sl2 / sw;
                float var = el2 - el * el;
                if (!(var > 0.0f)) var = 0.0f;
                var *= 4.0f / (float)max(h, 1);
                o = make_float4(scx / sw, scy / sw, scz / sw, var);
            }
            s_spatial = 1.0f;
        }
        a.v_color[i] = o;
        s_var = o.w; s_h = (float)h; s_n = 1.0f;
    }
    s[0] = s_var; s[1] = s_spatial; s[2] = s_h; s[3] = s_n;
}

// one workgroup per tile
template <int RFIX>
__global__ __launch_bounds__(256) void svgf_variance_kernel(VarianceArgs a)
{
    const int tile_y = a.row0 / 4 + blockIdx.y;
    if (a.tile_flags) {
        // whole tile long history: leave.  In the steady state that is ~98% of the 32k workgroups of a 4K
        // frame, so the test is a SCALAR load of the aligned word holding the flag (uniform address),
        // not a vector byte load: the launch is bound by how fast empty workgroups retire.
        const size_t fi = (size_t)tile_y * a.tiles_x + blockIdx.x;
        const unsigned word = reinterpret_cast<const unsigned*>(a.tile_flags)[fi >> 2];
        if (((word >> (8u * (unsigned)(fi & 3))) & 0xffu) == 0u) return;
    }
    float s[4];
    variance_pixel<RFIX>(a, blockIdx.x, tile_y, s);
    float s_var = s[0], s_spatial = s[1], s_h = s[2], s_n = s[3];
    if (a.stats) {
        // wavefront __shfl reductions, then the 4 waves of the workgroup through LDS, then ONE
        // atomic per workgroup and statistic (520k same-address atomics per 4K frame, one per
        // wave, cost 6 ms; 32k cost ~0.1 ms).  Optional diagnostics: off in the timed pipeline.
        __shared__ float part[4][4];
        s_var = wave_sum(s_var); s_spatial = wave_sum(s_spatial); s_h = wave_sum(s_h); s_n = wave_sum(s_n);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { part[wave][0] = s_var; part[wave][1] = s_spatial; part[wave][2] = s_h; part[wave][3] = s_n; }
        __syncthreads();
        if (threadIdx.x < 4) {
            const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
            if (part[0][3] + part[1][3] + part[2][3] + part[3][3] > 0.0f) atomicAdd(&a.stats[threadIdx.x], t);
        }
    }
}

// Fused frames (tile flags from T, v_color prefilled, no statistics), radius 3.  In the steady state ~2 % of the
// tiles are flagged (disocclusion bands), and the launch had two costs of ~40 us each at 4K, whichever form it
// took: one workgroup per tile is bound by launching and retiring 130 k waves that only read a flag; a few
// thousand workgroups walking the flags are bound by the flagged tile itself, a chain of 7 x 14 dependent
// gathers per pixel.  This kernel removes both: kVGrid workgroups (a prime, so the tiles of a vertical or a
// horizontal band land on different workgroups) each test every kVGrid-th flag with one load + ballot, and a
// flagged tile first stages its 70 x 10 neighbourhood of t_color and nd in LDS with six coalesced loads per
// thread, all in flight together; the 49 taps read LDS.  Same taps, order and arithmetic as variance_pixel:
// identical bits.  4K steady state (~750 flagged tiles, ~10 k short-history pixels): 38 -> 26 us, of which 4.5 are
// the empty launch, 2.4 the staging and the rest the tap bodies on the CUs that hold several flagged tiles;
// a frame after a scene cut (every tile flagged): 670 -> 590 us.
constexpr int kVGrid = 2039;
__global__ __launch_bounds__(256) void svgf_variance_tile_kernel(VarianceArgs a, int tiles_y)
{
    __shared__ float4 sc[kVH][kVW], sn[kVH][kVW];
    __shared__ int todo[256], wave_count[4];
    const Geom g = a.g;
    const int ntiles = a.tiles_x * tiles_y;
    const int lane = threadIdx.x & 63;
    const unsigned char* flags = a.tile_flags + (size_t)(a.row0 / 4) * a.tiles_x;
    // rows that exist in this buffer: the launcher checked that [row0 - 3, row1 + 3) clamped to the frame is inside
    const int blo = max(g.buf_row0, 0), bhi = min(g.buf_row0 + g.buf_rows, g.H);
    for (int base = 0; base < ntiles; base += 64 * (int)gridDim.x) {
        const int t = base + lane * (int)gridDim.x + (int)blockIdx.x;
        unsigned long long m = __builtin_amdgcn_ballot_w64(t < ntiles && flags[t] != 0);    // the same in all four waves
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            const int tile = base + k * (int)gridDim.x + (int)blockIdx.x;
            const int x0 = (tile % a.tiles_x) * 64, y0 = (a.row0 / 4 + tile / a.tiles_x) * 4;
            for (int q = threadIdx.x; q < kVW * kVH; q += 256) {
                const int ry = q / kVW, rx = q - ry * kVW;
                const int tx = min(max(x0 - kVR + rx, 0), g.W - 1), ty = min(max(y0 - kVR + ry, blo), bhi - 1);
                const size_t ti = pix_index(g, tx, ty);
                // a pixel of a tile T did not flag was not written to t_color (sparse_t_color); V never writes such
                // a pixel either, so v_color still holds T's value for it
                const float4* src = a.t_color;
                if (a.sparse_t_color && a.tile_flags[(size_t)(ty >> 2) * a.tiles_x + (tx >> 6)] == 0) src = a.v_color;
                sc[ry][rx] = src[ti];
                sn[ry][rx] = a.nd[ti];
            }
            // The short-history pixels of a flagged tile are few (a band a few pixels wide: ~13 of 256 in the steady
            // state), and a wave with one such lane pays for the whole 49-tap body: compact them to the first
            // threads of the workgroup, so that one wave does the work of four.
            int h = 0;
            {
                const int x = x0 + (threadIdx.x & 63), y = y0 + (threadIdx.x >> 6);
                const bool active = x < g.W && y >= a.row0 && y < a.row1;
                if (active) h = (int)a.t_len[pix_index(g, x, y)];
                const bool spatial = active && h < a.h_threshold;  // (else v_color already holds t_color)
                const unsigned long long b = __builtin_amdgcn_ballot_w64(spatial);
                const int wave = threadIdx.x >> 6;
                if (lane == 0) wave_count[wave] = __builtin_popcountll(b);
                __syncthreads();                                   // also: the staged region is complete
                int pos = __builtin_popcountll(b & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; ++w) pos += wave_count[w];
                if (spatial) todo[pos] = (int)threadIdx.x | (h << 8);
                __syncthreads();
            }
            const int ntodo = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            if ((int)threadIdx.x < ntodo) {
                const int id = todo[threadIdx.x];
                const int lx = id & 63, ly = (id >> 6) & 3;
                h = id >> 8;
                const int x = x0 + lx, y = y0 + ly;
                bool keep;                                         // (weights vanished: o = sc's centre, variance included)
                const float4 o = variance_window_lds([&](int ry, int rx) { return sc[ry][rx]; }, [&](int ry, int rx) { return sn[ry][rx]; },
                                                     lx, ly, x, y, g, a.sigma_n, a.sigma_z, h, keep);
                a.v_color[pix_index(g, x, y)] = o;
            }
            __syncthreads();                                       // the next flagged tile restages the LDS region
        }
    }
}

#ifdef RMD_EXPERIMENTS
// Fused frames only (tile flags from T, no statistics): a fixed number of workgroups, each looking
// at the flags of every gridDim.x-th tile (64 per coalesced-by-stride load and ballot) and running
// the flagged ones.  Constant footprint, for the same reason as svgf_temporal_persistent_kernel; the
// stride spreads runs of flagged tiles (a disoccluded band along a frame edge) over the workgroups.
template <int RFIX>
__global__ __launch_bounds__(256) void svgf_variance_persistent_kernel(VarianceArgs a, int tiles_y)
{
    __builtin_amdgcn_s_setprio(3);
    const int ntiles = a.tiles_x * tiles_y;
    const int lane = threadIdx.x & 63;
    const unsigned char* flags = a.tile_flags + (size_t)(a.row0 / 4) * a.tiles_x;
    for (int base = 0; base < ntiles; base += 64 * (int)gridDim.x) {
        const int t = base + lane * (int)gridDim.x + (int)blockIdx.x;
        unsigned long long m = __builtin_amdgcn_ballot_w64(t < ntiles && flags[t] != 0);    // the same in all four waves
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            const int tile = base + k * (int)gridDim.x + (int)blockIdx.x;
            float s[4];
            variance_pixel<RFIX>(a, tile % a.tiles_x, a.row0 / 4 + tile / a.tiles_x, s);
        }
    }
}
#endif

}  // namespace rmd

using namespace rmd;

static int env_v_workgroups()
{
    static const int v = tuning_env("RMD_V_WORKGROUPS", -1);
    return v;
}

// true when a fused frame's V pass will be svgf_variance_tile_kernel, the one form that can take the unflagged pixels of
// its windows from v_color: rmd_svgf_frame_tv evaluates this ONCE and hands the answer to both launchers, so T (which
// then writes t_color only inside flagged tiles) and V cannot disagree about it
bool rmd::variance_reads_sparse_t_color(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, bool fused)
{
    static const int sparse = tuning_env("RMD_SPARSE_T_COLOR", 1);
    return sparse && fused && !f->stats && f->v_tile_flags && p->var_radius == kVR && p->tv_workgroups == 0 && env_v_workgroups() < 0;
}

int rmd::launch_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream, bool fused,
                         bool sparse_t_color)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_variance: params is NULL");
    if (!f->t_color || !f->t_len || !f->nd || !f->v_color) return fail(RMD_E_NULL, "rmd_svgf_variance: a required plane is NULL");
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_variance: rows [%d,%d) invalid", row0, row1);
    if (p->var_radius < 0 || p->var_radius > 16) return fail(RMD_E_PARAM, "rmd_svgf_variance: var_radius %d outside [0,16]", p->var_radius);
    const int reach = p->var_radius > 1 ? p->var_radius : 1;
    if (int e = check_rows_in_buffer(f, row0 - reach, row1 + reach, "rmd_svgf_variance")) return e;
    const void* planes16[] = { f->t_color, f->nd, f->v_color };
    for (const void* q : planes16)
        if (!aligned_to(q, 16)) return fail(RMD_E_ALIGN, "rmd_svgf_variance: float4 planes must be 16-byte aligned");
    if (f->t_color == f->v_color) return fail(RMD_E_BUFFER, "rmd_svgf_variance: t_color and v_color alias");
    if (fused && f->v_tile_flags && !aligned_to(f->v_tile_flags, 4)) return fail(RMD_E_ALIGN, "rmd_svgf_variance: v_tile_flags must be 4-byte aligned");

    VarianceArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.t_color = (const float4*)f->t_color; a.t_len = f->t_len; a.nd = (const float4*)f->nd;
    a.v_color = (float4*)f->v_color; a.stats = f->stats;
    a.row0 = row0; a.row1 = row1;
    a.h_threshold = p->var_h_threshold; a.radius = p->var_radius;
    a.prefilled = (fused && !f->stats) ? 1 : 0;
    a.tile_flags = a.prefilled ? f->v_tile_flags : nullptr;
    a.tiles_x = (f->width + 63) / 64;
    a.sigma_n = p->sigma_n; a.sigma_z = p->sigma_z;
    dim3 grid((f->width + 63) / 64, (row1 - 1) / 4 - row0 / 4 + 1);
    if (p->tv_workgroups < 0 || p->tv_workgroups > 65536) return fail(RMD_E_PARAM, "rmd_svgf_variance: tv_workgroups %d outside [0,65536]", p->tv_workgroups);
    const int v_wgs = p->tv_workgroups > 0 ? p->tv_workgroups : env_v_workgroups();
    a.sparse_t_color = sparse_t_color ? 1 : 0;
    const bool tile_form = a.tile_flags && a.radius == kVR && v_wgs < 0;
    // T wrote t_color only inside the tiles it flagged: every other form of V would read stale pixels
    if (sparse_t_color && !tile_form)
        return fail(RMD_E_PARAM, "rmd_svgf_variance: sparse t_color needs the tile form of V (fused frame, tile flags, var_radius %d, no statistics)", kVR);
    if (a.tile_flags && v_wgs > 0) {
#ifdef RMD_EXPERIMENTS
        const dim3 pg(v_wgs);
        if (a.radius == 3) hipLaunchKernelGGL(svgf_variance_persistent_kernel<3>, pg, dim3(256), 0, as_stream(stream), a, (int)grid.y);
        else               hipLaunchKernelGGL(svgf_variance_persistent_kernel<0>, pg, dim3(256), 0, as_stream(stream), a, (int)grid.y);
#else
        return fail(RMD_E_UNSUPPORTED, "rmd_svgf_variance: tv_workgroups > 0 (persistent T / V grids) is an experiment (make experiments)");
#endif
    } else if (tile_form) {
        hipLaunchKernelGGL(svgf_variance_tile_kernel, dim3(kVGrid), dim3(256), 0, as_stream(stream), a, (int)grid.y);
    } else {
        if (a.radius == 3) hipLaunchKernelGGL(svgf_variance_kernel<3>, grid, dim3(256), 0, as_stream(stream), a);
        else               hipLaunchKernelGGL(svgf_variance_kernel<0>, grid, dim3(256), 0, as_stream(stream), a);
    }
    RMD_LAUNCH_CHECK("svgf_variance_kernel");
    return RMD_OK;
}

extern "C" int rmd_svgf_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream)
{
    return rmd::launch_variance(f, p, row0, row1, stream, /*fused=*/false, /*sparse_t_color=*/false);
}
