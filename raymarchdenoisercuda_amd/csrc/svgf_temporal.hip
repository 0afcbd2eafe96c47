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
#include "svgf_tv.h"

namespace rmd {

// Whole frames (rmd_svgf_frame_tv): T and V in ONE launch.  V rewrites only the pixels with a short history (~10 k of 8.3 M
// at 4K in the steady state, in ~2 % of the tiles), but as a launch of its own it cost 26-30 us at EVERY frame size -- an empty
// grid, a flag scan, a dependent chain of loads per flagged tile (10 % of a 1080p frame).  Here the workgroup that finds a
// short-history pixel in its tile finishes the job itself: V needs T's output on the tile + 3 pixels around it, which other
// workgroups own and may not have written yet, so it RECOMPUTES T for those 444 halo pixels (same code, same bits; 2 % of the
// tiles x 2.7 = +4 % of T's arithmetic), stages tile + halo in LDS and runs the 49 taps for its short-history pixels, compacted
// to the first lanes.  No t_color plane, no second launch; v_color gets T's value or V's, t_moments and the tile flags as before.
struct FusedVArgs {
    int row0, row1;           // rows V is responsible for (a strip: fewer than T's, whose output V's windows tap)
    int h_threshold;
    float sigma_n, sigma_z;
    int tiles_x, tiles_y;     // the launch's tile grid (one workgroup per tile, dealt out edges first)
#ifdef RMD_EXPERIMENTS
    int xcd_group;            // > 0: the interior tiles in XCD-owned column groups of this many tile columns (below; measured slower); 0: row by row
#endif
};

// IN8 (rmd_svgf_gbuffer_frame): the same launch with the 8-bit front end of temporal_pixel<true> -- the workgroup reads the
// GBuffer's uchar4 render / albedo / normal planes (12 B/px instead of 32 B/px of float planes) and writes the float nd plane.
template <bool IN8>
__global__ __launch_bounds__(256) void svgf_temporal_variance_kernel(TemporalArgs a, FusedVArgs v)
{
    // 20 144 B of LDS: eight workgroups per CU, the occupancy of the plain T kernel (the pass is latency-bound: with two
    // float4 planes, 23 KB, it ran at seven)
    __shared__ float4 sn[kVH][kVW];
    __shared__ float scr[kVH][kVW], scg[kVH][kVW], scb[kVH][kVW];
    __shared__ unsigned short todo[256];
    __shared__ unsigned long long wave_mask[4];
    const Geom g = a.g;
    // Tiles are dealt out edges first (bottom row, top row, left column, right column, then the interior row by row): the
    // tiles with short-history pixels sit along the frame edges the camera moves away from, and their workgroups live
    // 3-4 x longer than the others -- at the end of the grid they would be the launch's tail.
    int bx, by;
    {
        const int nx = v.tiles_x, ny = v.tiles_y, id = (int)blockIdx.x;
        if (nx < 3 || ny < 3)               { bx = id % nx; by = id / nx; }
        else if (id < nx)                   { bx = id; by = ny - 1; }
        else if (id < 2 * nx)               { bx = id - nx; by = 0; }
        else if (id < 2 * nx + (ny - 2))    { bx = 0; by = 1 + id - 2 * nx; }
        else if (id < 2 * nx + 2 * (ny - 2)) { bx = nx - 1; by = 1 + id - 2 * nx - (ny - 2); }
#ifndef RMD_EXPERIMENTS
        else { const int k = id - 2 * nx - 2 * (ny - 2); bx = 1 + k % (nx - 2); by = 1 + k / (nx - 2); }
#else
        else if (v.xcd_group <= 0) { const int k = id - 2 * nx - 2 * (ny - 2); bx = 1 + k % (nx - 2); by = 1 + k / (nx - 2); }
        else {
            // EXPERIMENTS BUILD (RMD_TV_XCD_GROUP=G; measured and lost, DESIGN.md section 4.6: 4K 169-170 us row by row against 183-199 for
            // G = 1 ... 16, tools/tv_probe.py).  The interior, XCD-aware.  A 64x4 tile shares rows with the tiles above and below it: its reprojection taps reach one
            // history row beyond its own four, its depth gradient one nd row -- a quarter of the 41 + 16 bytes per pixel of those
            // planes.  Dealt out row by row, vertical neighbours are nx ids apart and land on different XCDs (workgroup id % 8), so
            // every shared row is fetched twice from the fabric: the counters show 894 MB read per 4K launch against 672
            // algorithmic (profiles/r04_pmc_frame.txt).  Here the interior is ONE sequence Q in column-group-major order (groups of
            // xcd_group tile columns, row-major inside a group), cut into eight contiguous runs of (almost) equal length, one per
            // XCD: the workgroups id, id + 8, id + 16 ... of an XCD walk down ITS run, so a tile and the tile below it run on the
            // same XCD xcd_group positions apart and the shared rows hit its L2.  Every XCD has the same number of tiles (no tail), and a
            // group row is xcd_group KB of contiguous addresses per plane row.
            const int first = 2 * nx + 2 * (ny - 2), total = nx * ny, wi = nx - 2, hi = ny - 2;
            const int xcd = id & (kXcds - 1);
            int start = 0, len = 0;
            for (int j = 0; j < kXcds; ++j) {                       // interior ids with residue j: first_j, first_j + 8, ...
                const int first_j = first + ((j - first) & (kXcds - 1));
                const int len_j = first_j < total ? (total - first_j + kXcds - 1) / kXcds : 0;
                if (j < xcd) start += len_j;
                if (j == xcd) len = len_j;
            }
            const int first_x = first + ((xcd - first) & (kXcds - 1));
            const int q = start + (id - first_x) / kXcds;           // position in Q
            const int G = v.xcd_group, per_group = G * hi;
            const int grp = q / per_group, r = q - grp * per_group;
            const int gw = min(G, wi - grp * G);                    // (the last group may be narrower)
            bx = 1 + grp * G + r % gw;
            by = 1 + r / gw;
        }
#endif
    }
    const int tile_x = bx, tile_y = a.row0 / 4 + by;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x0 = tile_x * 64, y0 = tile_y * 4;
    const int x = x0 + lx, y = y0 + ly;
    const bool active = x < g.W && y >= a.row0 && y < a.row1;
    float4 tc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nd_own = tc;
    int h = 0;
    bool spatial = false;
    if (active) {
        const size_t i = pix_index(g, x, y);
        float2 mom;
        int4 dbg;
        temporal_pixel<IN8>(a, x, y, tc, mom, dbg, nd_own);
        if constexpr (IN8) a.nd_out[i] = nd_own;
        a.t_moments[i] = mom;
        a.t_len[i] = (unsigned char)dbg.w;
        if (a.t_debug) a.t_debug[i] = dbg;
        h = dbg.w;
        spatial = h < v.h_threshold && y >= v.row0 && y < v.row1;
        if (!spatial) a.v_color[i] = tc;          // (a short-history pixel gets its value below, from another lane)
    }
    const unsigned long long mine = __builtin_amdgcn_ballot_w64(spatial);     // bit = column, wave = row of the tile
    if (lx == 0) wave_mask[ly] = mine;
    const int any = __syncthreads_or(spatial ? 1 : 0);
    if (a.tile_flags && threadIdx.x == 0) a.tile_flags[(size_t)tile_y * a.tiles_x + tile_x] = (unsigned char)(any != 0);
    if (!any) return;

    // ---- T's output and nd on the tile + halo into LDS.  Own pixel from registers; the halo pixels are RECOMPUTED, and only those
    // some 7x7 window of a short-history pixel reaches (Chebyshev distance <= 3 to such a pixel), dealt out to the threads without
    // holes: a silhouette that crosses the tile needs ~100 of the 444 halo pixels, and as the bounding box of round 3 they were spread
    // over two rounds of the recomputation (three dependent memory round trips each) in which most lanes idled.
    // Everything that decides WHICH cells is wave-uniform (the four row masks of the tile): per staged row ry the 70-bit mask of the
    // needed cells (need_lo: columns 0..63 of the staged region, need_hi: 64..69) and the running count, in scalar registers.
    const unsigned long long m0 = wave_mask[0], m1 = wave_mask[1], m2 = wave_mask[2], m3 = wave_mask[3];
    unsigned long long need_lo[kVH];
    unsigned need_hi[kVH];
    int before[kVH + 1];
    before[0] = 0;
#pragma unroll
    for (int ry = 0; ry < kVH; ++ry) {
        // tile rows r with |r - (ry - 3)| <= 3, i.e. ry - 6 <= r <= ry
        unsigned long long rows = 0ull;
        if (ry <= 6) rows |= m0;
        if (ry >= 1 && ry <= 7) rows |= m1;
        if (ry >= 2 && ry <= 8) rows |= m2;
        if (ry >= 3) rows |= m3;
        // a pixel in tile column c reaches staged columns c .. c + 6
        unsigned long long lo = rows;
        unsigned hi = 0u;
#pragma unroll
        for (int d = 1; d <= 2 * kVR; ++d) { lo |= rows << d; hi |= (unsigned)(rows >> (64 - d)); }
        if (ry >= kVR && ry < kVR + 4) { lo &= (1ull << kVR) - 1ull; hi &= ~((1u << kVR) - 1u); }   // the tile itself: columns 3 .. 66
        // pixels outside the frame are never tapped (A.V skips them); neither are rows outside T's range: V's rows lie
        // at least 3 rows inside it wherever the frame goes on (launch_temporal_variance)
        const int ty = y0 - kVR + ry;
        if (ty < a.row0 || ty >= a.row1) { lo = 0ull; hi = 0u; }
        if (x0 == 0) lo &= ~((1ull << kVR) - 1ull);                                     // staged columns 0 .. 2 are left of the frame
        const int cols_in = g.W - (x0 - kVR);                                           // staged columns < cols_in are inside the frame
        if (cols_in < kVW) { lo &= cols_in >= 64 ? ~0ull : ((1ull << cols_in) - 1ull); hi &= cols_in > 64 ? ((1u << (cols_in - 64)) - 1u) : 0u; }
        need_lo[ry] = lo; need_hi[ry] = hi & 0x3fu;
        before[ry + 1] = before[ry] + __builtin_popcountll(lo) + __builtin_popcount(hi & 0x3fu);
    }
    if (active) {
        scr[ly + kVR][lx + kVR] = tc.x; scg[ly + kVR][lx + kVR] = tc.y; scb[ly + kVR][lx + kVR] = tc.z;
        if constexpr (IN8) sn[ly + kVR][lx + kVR] = nd_own;
        else sn[ly + kVR][lx + kVR] = a.nd[pix_index(g, x, y)];
    }
    constexpr int kRing = kVW * kVH - 64 * 4;                    // 444: the whole halo ring
    const bool whole_ring = before[kVH] == kRing;                // frames without history: every tile, every cell -- no need to search
    for (int i = threadIdx.x; i < before[kVH]; i += 256) {
        int ry, rx;
        if (whole_ring) {                                        // (workgroup-uniform) rows 0..2 | the six side columns of rows 3..6 | rows 7..9
            if (i < kVR * kVW)               { ry = i / kVW; rx = i - ry * kVW; }
            else if (i < kVR * kVW + 4 * 6)  { const int j = i - kVR * kVW; ry = kVR + j / 6; const int c = j - (j / 6) * 6; rx = c < kVR ? c : 64 + c; }
            else                             { const int j = i - kVR * kVW - 4 * 6; ry = kVR + 4 + j / kVW; rx = j - (j / kVW) * kVW; }
        } else {
            // the i-th needed cell: its row by the running counts, then the (i - before[ry])-th set bit of the row's mask
            int j = i;
            ry = 0;
            unsigned long long lo = need_lo[0];
            unsigned hi = need_hi[0];
#pragma unroll
            for (int r = 1; r < kVH; ++r)
                if (i >= before[r]) { ry = r; j = i - before[r]; lo = need_lo[r]; hi = need_hi[r]; }
            rx = 0;                                                                      // binary search: set bits below rx <= j
#pragma unroll
            for (int step = 64; step >= 1; step >>= 1) {
                const int cand = rx + step;
                const int below = cand >= 64 ? __builtin_popcountll(lo) + __builtin_popcount(hi & ((1u << min(cand - 64, 31)) - 1u))
                                             : __builtin_popcountll(lo & ((1ull << cand) - 1ull));
                if (cand < kVW && below <= j) rx = cand;
            }
        }
        const int tx = x0 - kVR + rx, ty = y0 - kVR + ry;
        float4 hc, hn;
        float2 mom;
        int4 dbg;
        temporal_pixel<IN8>(a, tx, ty, hc, mom, dbg, hn);
        scr[ry][rx] = hc.x; scg[ry][rx] = hc.y; scb[ry][rx] = hc.z;
        if constexpr (IN8) sn[ry][rx] = hn;
        else sn[ry][rx] = a.nd[pix_index(g, tx, ty)];
    }
    // ---- the short-history pixels of the tile, compacted to the first lanes (a wave with one such lane pays for the whole
    // 49-tap body), as svgf_variance_tile_kernel does
    const int c0 = __builtin_popcountll(m0), c1 = __builtin_popcountll(m1), c2 = __builtin_popcountll(m2), c3 = __builtin_popcountll(m3);
    if (spatial) {
        const int before = ly == 0 ? 0 : ly == 1 ? c0 : ly == 2 ? c0 + c1 : c0 + c1 + c2;
        todo[before + __builtin_popcountll(mine & ((1ull << lx) - 1ull))] = (unsigned short)(threadIdx.x | (h << 8));
    }
    __syncthreads();                                       // todo[] and the staged region are complete
    if ((int)threadIdx.x < c0 + c1 + c2 + c3) {
        const int id = todo[threadIdx.x];
        const int px = id & 63, py = (id >> 6) & 3;
        bool keep;
        float4 o = variance_window_lds([&](int ry, int rx) { return make_float4(scr[ry][rx], scg[ry][rx], scb[ry][rx], 0.0f); },
                                       [&](int ry, int rx) { return sn[ry][rx]; }, px, py, x0 + px, y0 + py, g, v.sigma_n, v.sigma_z, id >> 8, keep);
        if (keep) {                                        // weights vanished (normals that break the unit-length contract): the pixel
            float4 kn;                                     // keeps T's value, whose variance only the lane that computed it had
            float2 mom;
            int4 dbg;
            temporal_pixel<IN8>(a, x0 + px, y0 + py, o, mom, dbg, kn);
        }
        a.v_color[pix_index(g, x0 + px, y0 + py)] = o;
    }
}

// one workgroup per tile: the pass on its own runs at HBM speed this way
__global__ __launch_bounds__(256) void svgf_temporal_kernel(TemporalArgs a)
{
    temporal_tile(a, blockIdx.x, a.row0 / 4 + blockIdx.y);
}

#ifdef RMD_EXPERIMENTS
// Tiles claimed from a device counter shared with the a-trous launches that carried the pass as a side job
__global__ __launch_bounds__(256) void svgf_temporal_claim_kernel(TemporalArgs a, unsigned* counter, int units)
{
    __shared__ unsigned slot;
    for (;;) {
        if (threadIdx.x == 0) slot = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned u = slot;
        if (u >= (unsigned)units) return;                 // workgroup-uniform
        temporal_tile(a, (int)(u % (unsigned)a.tiles_x), a.row0 / 4 + (int)(u / (unsigned)a.tiles_x));
        __syncthreads();                                  // everybody has read the slot (a pass without tile flags has no barrier of its own)
    }
}

#endif

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

int rmd::make_temporal_args(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, bool fused, bool sparse_t_color,
                            TemporalArgs* out)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_temporal: params is NULL");
    if (!f->color || !f->nd || !f->motion || !f->t_color || !f->t_moments || !f->t_len)
        return fail(RMD_E_NULL, "rmd_svgf_temporal: a required plane is NULL");
    // hist_color / hist_moments / hist_len / prev_nd all NULL = "no history": every pixel is a disocclusion
    const bool has_hist = f->hist_color && f->hist_moments && f->hist_len && f->prev_nd;
    if (!has_hist && (f->hist_color || f->hist_moments || f->hist_len || f->prev_nd))
        return fail(RMD_E_NULL, "rmd_svgf_temporal: history planes must be all set or all NULL");
    if (row0 < 0 || row1 > f->height || row0 >= row1) return fail(RMD_E_ROWS, "rmd_svgf_temporal: rows [%d,%d) invalid", row0, row1);
    if (p->max_motion_rows < 0 || p->h_max < 1 || p->h_max > 255) return fail(RMD_E_PARAM, "rmd_svgf_temporal: max_motion_rows < 0 or h_max outside [1,255]");
    if (p->tv_workgroups < 0 || p->tv_workgroups > 65536) return fail(RMD_E_PARAM, "rmd_svgf_temporal: tv_workgroups %d outside [0,65536]", p->tv_workgroups);
    // current-frame planes: +1 row (depth gradient); history planes: +-(max_motion_rows) rows
    if (int e = check_rows_in_buffer(f, row0, row1 + 1, "rmd_svgf_temporal (current frame)")) return e;
    if (has_hist)
        if (int e = check_rows_in_buffer(f, row0 - p->max_motion_rows, row1 + p->max_motion_rows, "rmd_svgf_temporal (history)")) return e;
    const void* planes16[] = { f->color, f->nd, f->hist_color, f->prev_nd, f->t_color, f->t_debug };
    for (const void* q : planes16)
        if (!aligned_to(q, 16)) return fail(RMD_E_ALIGN, "rmd_svgf_temporal: float4 planes must be 16-byte aligned");
    if (!aligned_to(f->motion, 8) || !aligned_to(f->hist_moments, 8) || !aligned_to(f->t_moments, 8))
        return fail(RMD_E_ALIGN, "rmd_svgf_temporal: float2 planes (motion, moments) must be 8-byte aligned");

    TemporalArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.color = (const float4*)f->color; a.nd = (const float4*)f->nd; a.motion = (const float2*)f->motion;
    a.hist_color = (const float4*)f->hist_color; a.hist_moments = (const float2*)f->hist_moments; a.hist_len = f->hist_len;
    a.prev_nd = (const float4*)f->prev_nd;
    a.t_color = (float4*)f->t_color; a.t_moments = (float2*)f->t_moments; a.t_len = f->t_len; a.t_debug = (int4*)f->t_debug;
    a.v_color = nullptr;
    a.render8 = a.albedo8 = a.normal8 = nullptr; a.nd_out = nullptr; a.albedo_eps = 0.0f;
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
    *out = a;
    return RMD_OK;
}

int rmd::launch_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream, bool fused,
                         bool sparse_t_color)
{
    TemporalArgs a;
    if (int e = make_temporal_args(f, p, row0, row1, fused, sparse_t_color, &a)) return e;
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

#ifdef RMD_EXPERIMENTS
// What the a-trous launches did not get to (rmd_svgf_frame_atrous_next): every workgroup claims tiles from the same counter
// until it runs out.  Nobody waits for anybody; a launch that finds the counter at `units` returns at once.
int rmd::launch_temporal_claim(const AtrousSide& side, void* stream)
{
    if (side.units <= 0) return RMD_OK;
    const int wgs = side.units < 8 * device_cus() ? side.units : 8 * device_cus();
    hipLaunchKernelGGL(svgf_temporal_claim_kernel, dim3(wgs), dim3(256), 0, as_stream(stream), side.t, side.counter, side.units);
    RMD_LAUNCH_CHECK("svgf_temporal_claim_kernel");
    return RMD_OK;
}
#endif

// T + V of a whole frame in one launch (svgf_temporal_variance_kernel): T on rows [row0,row1), V on [v_row0,v_row1).
int rmd::launch_temporal_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, int v_row0, int v_row1, void* stream,
                                  const GBuffer8* g8)
{
    if (int e = check_frame_geometry(f)) return e;
    if (!p) return fail(RMD_E_NULL, "rmd_svgf_frame_tv: params is NULL");
    // 8-bit front end: render / albedo / normal are the inputs, f->nd is WRITTEN, f->color is not used, f->motion may be NULL
    if (g8) {
        if (!g8->render || !g8->albedo || !g8->normal || !f->nd || !f->t_moments || !f->t_len || !f->v_color)
            return fail(RMD_E_NULL, "rmd_svgf_gbuffer_frame: a required plane is NULL");
        if (!aligned_to(g8->render, 4) || !aligned_to(g8->albedo, 4) || !aligned_to(g8->normal, 4))
            return fail(RMD_E_ALIGN, "rmd_svgf_gbuffer_frame: uchar4 planes must be 4-byte aligned");
        if (!(g8->albedo_eps > 0.0f)) return fail(RMD_E_PARAM, "rmd_svgf_gbuffer_frame: albedo_eps must be > 0");
        if (f->nd == f->prev_nd) return fail(RMD_E_BUFFER, "rmd_svgf_gbuffer_frame: the nd plane written aliases prev_nd");
    } else if (!f->color || !f->nd || !f->motion || !f->t_moments || !f->t_len || !f->v_color)
        return fail(RMD_E_NULL, "rmd_svgf_frame_tv: a required plane is NULL");
    const bool has_hist = f->hist_color && f->hist_moments && f->hist_len && f->prev_nd;
    if (!has_hist && (f->hist_color || f->hist_moments || f->hist_len || f->prev_nd))
        return fail(RMD_E_NULL, "rmd_svgf_frame_tv: history planes must be all set or all NULL");
    if (row0 < 0 || row1 > f->height || row0 >= row1 || v_row0 < row0 || v_row1 > row1 || v_row0 >= v_row1)
        return fail(RMD_E_ROWS, "rmd_svgf_frame_tv: rows T [%d,%d) V [%d,%d) invalid", row0, row1, v_row0, v_row1);
    // V's windows must find T's output (recomputed in the kernel) inside T's rows wherever the frame goes on
    if ((v_row0 - kVR < row0 && row0 > 0) || (v_row1 + kVR > row1 && row1 < f->height))
        return fail(RMD_E_ROWS, "rmd_svgf_frame_tv: V rows [%d,%d) need T on %d more rows than [%d,%d)", v_row0, v_row1, kVR, row0, row1);
    if (p->var_radius != kVR) return fail(RMD_E_PARAM, "rmd_svgf_frame_tv: the fused kernel is built for var_radius %d", kVR);
    if (p->max_motion_rows < 0 || p->h_max < 1 || p->h_max > 255) return fail(RMD_E_PARAM, "rmd_svgf_frame_tv: max_motion_rows < 0 or h_max outside [1,255]");
    if (int e = check_rows_in_buffer(f, row0, row1 + 1, "rmd_svgf_frame_tv (current frame)")) return e;
    if (has_hist)
        if (int e = check_rows_in_buffer(f, row0 - p->max_motion_rows, row1 + p->max_motion_rows, "rmd_svgf_frame_tv (history)")) return e;
    const void* planes16[] = { g8 ? nullptr : f->color, f->nd, f->hist_color, f->prev_nd, f->v_color, f->t_debug };
    for (const void* q : planes16)
        if (!aligned_to(q, 16)) return fail(RMD_E_ALIGN, "rmd_svgf_frame_tv: float4 planes must be 16-byte aligned");
    if (!aligned_to(f->motion, 8) || !aligned_to(f->hist_moments, 8) || !aligned_to(f->t_moments, 8))
        return fail(RMD_E_ALIGN, "rmd_svgf_frame_tv: float2 planes (motion, moments) must be 8-byte aligned");

    TemporalArgs a;
    a.g = Geom{ f->width, f->height, f->buf_row0, f->buf_rows };
    a.color = (const float4*)f->color; a.nd = (const float4*)f->nd; a.motion = (const float2*)f->motion;
    a.render8 = a.albedo8 = a.normal8 = nullptr; a.nd_out = nullptr; a.albedo_eps = 0.0f;
    if (g8) {
        a.color = nullptr; a.nd = nullptr;
        a.render8 = g8->render; a.albedo8 = g8->albedo; a.normal8 = g8->normal; a.nd_out = (float4*)f->nd; a.albedo_eps = g8->albedo_eps;
    }
    a.hist_color = (const float4*)f->hist_color; a.hist_moments = (const float2*)f->hist_moments; a.hist_len = f->hist_len;
    a.prev_nd = (const float4*)f->prev_nd;
    a.t_color = nullptr; a.t_moments = (float2*)f->t_moments; a.t_len = f->t_len; a.t_debug = (int4*)f->t_debug;
    a.v_color = (float4*)f->v_color;
    a.sparse_t_color = 0;
    a.tile_flags = f->v_tile_flags; a.tiles_x = (f->width + 63) / 64; a.var_h_threshold = p->var_h_threshold;
    a.row0 = row0; a.row1 = row1;
    a.alpha_color = p->alpha_color; a.alpha_moments = p->alpha_moments; a.k_z = p->k_z; a.k_n = p->k_n;
    a.h_max = p->h_max; a.max_motion_rows = p->max_motion_rows;
    const int tiles_x = (f->width + 63) / 64, tiles_y = (row1 - 1) / 4 - row0 / 4 + 1;
    FusedVArgs v = { v_row0, v_row1, p->var_h_threshold, p->sigma_n, p->sigma_z, tiles_x, tiles_y };
#ifdef RMD_EXPERIMENTS
    v.xcd_group = tuning_env("RMD_TV_XCD_GROUP", 0);       // A/B knob of tools/tv_probe.py
#endif
    const dim3 grid((unsigned)tiles_x * (unsigned)tiles_y);
    if (g8) hipLaunchKernelGGL(HIP_KERNEL_NAME(svgf_temporal_variance_kernel<true>), grid, dim3(256), 0, as_stream(stream), a, v);
    else    hipLaunchKernelGGL(HIP_KERNEL_NAME(svgf_temporal_variance_kernel<false>), grid, dim3(256), 0, as_stream(stream), a, v);
    RMD_LAUNCH_CHECK("svgf_temporal_variance_kernel");
    return RMD_OK;
}

extern "C" int rmd_svgf_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream)
{
    return rmd::launch_temporal(f, p, row0, row1, stream, /*fused=*/false, /*sparse_t_color=*/false);
}
