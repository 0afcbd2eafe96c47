// box_filter.hip — the uchar4 filter the reference actually implements, written for gfx950.
//
// Parity target: reference src/filter.cu:13-58 (filterKernelBaseline) and :87-158
// (filterKernelTiled with cacheInput=false): mean of the in-bounds taps of a (2r+1)^2 window,
// fp32 accumulate, one division, truncating cast.  Bit-exact (SURVEY §8c known answers).
//
// MI355X design (not the reference's 16x16 CUDA blocks):
//   - workgroup = 256 threads = 4 wave64s; a wave owns 64 consecutive x, so each row access of a
//     wave is one 256-byte coalesced uchar4 segment.
//   - direct kernel: float accumulation in the reference's tap order (dx outer, dy inner).
//   - stream kernel (radius 1..4, the reference's radius 2): no LDS, DPP neighbours, a register ring of row sums.
//   - scan kernel (every other radius up to 32; the job of reference cacheTile src/filter.cu:60-85: tile + halo
//     staged once per workgroup): prefix sums along the staged rows, running sums down the columns; the staged form
//     is the rows' window sums in LDS with ONE consistent stride (the reference stages with a rounded stride and
//     reads with the unrounded one).  The sums are integers < 2^24, so they equal the reference's float
//     accumulation exactly, for any order.
//   - LDS kernel (experiments build; what the scan kernel replaced): 64x16 tile + halo in LDS, 2r+1 taps per sum.
//   - one launch per level: the reference's in-kernel level loop has only __syncthreads()
//     between levels although taps cross blocks (src/filter.cu:56,156) — an inter-block race.
#include "common.h"

namespace rmd {

constexpr int kBoxBlockX = 64;   // one wave wide
constexpr int kBoxBlockY = 4;    // 4 waves
constexpr int kTileY     = 16;   // output rows per workgroup in the LDS kernel
constexpr int kMaxExactRadius = 127;   // (2r+1)^2 * 255 < 2^24

// ---- direct kernel: reference tap order, float accumulators ---------------------------------
template <bool GRAY>
__global__ __launch_bounds__(256) void box_direct_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out,
                                                         int W, int H, int radius)
{
    const int x = blockIdx.x * kBoxBlockX + (threadIdx.x & 63);
    const int y = blockIdx.y * kBoxBlockY + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float ax = 0.0f, ay = 0.0f, az = 0.0f, norm = 0.0f;
    for (int dx = -radius; dx <= radius; ++dx) {
        const int nx = x + dx;
        if (nx < 0 || nx >= W) continue;
        for (int dy = -radius; dy <= radius; ++dy) {
            const int ny = y + dy;
            if (ny < 0 || ny >= H) continue;
            const uchar4 t = in[(size_t)ny * W + nx];
            ax += (float)t.x;
            if (!GRAY) { ay += (float)t.y; az += (float)t.z; }
            norm += 1.0f;
        }
    }
    uchar4 o;
    if (GRAY) {
        // reference src/filter.cu:51-53: all three channels take the R mean
        const unsigned char g = (unsigned char)(ax / norm);
        o = make_uchar4(g, g, g, 0);
    } else {
        o = make_uchar4((unsigned char)(ax / norm), (unsigned char)(ay / norm), (unsigned char)(az / norm), 0);
    }
    out[(size_t)y * W + x] = o;
}

#ifdef RMD_EXPERIMENTS
// ---- LDS kernel: tile + halo staged once, separable integer box sum -------------------------
// dynamic LDS: uchar4 tile[(kTileY+2r)][64+2r] followed by uint2 hsum[(kTileY+2r)][64]
// (hsum.x = R | G<<16, hsum.y = B; each row sum <= 255*(2r+1) <= 65025 fits 16 bits)
template <bool GRAY>
__global__ __launch_bounds__(256) void box_lds_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out,
                                                      int W, int H, int radius)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char box_lds[];
    const int tileW = kBoxBlockX + 2 * radius;
    const int tileH = kTileY + 2 * radius;
    uchar4* tile = reinterpret_cast<uchar4*>(box_lds);
    uint2* hsum = reinterpret_cast<uint2*>(box_lds + (((size_t)tileW * tileH * sizeof(uchar4) + 15) & ~(size_t)15));

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * kBoxBlockX;
    const int y0 = blockIdx.y * kTileY;

    // 1. stage tile + halo; out-of-frame cells hold 0 so they add nothing to the sums
    for (int ty = wave; ty < tileH; ty += 4) {
        const int gy = y0 - radius + ty;
        const bool rowok = gy >= 0 && gy < H;
        for (int tx = lane; tx < tileW; tx += 64) {
            const int gx = x0 - radius + tx;
            uchar4 v = make_uchar4(0, 0, 0, 0);
            if (rowok && gx >= 0 && gx < W) v = in[(size_t)gy * W + gx];
            tile[ty * tileW + tx] = v;
        }
    }
    __syncthreads();

    // 2. horizontal sums over [tx, tx+2r] for the 64 output columns of every staged row
    for (int ty = wave; ty < tileH; ty += 4) {
        unsigned sr = 0, sg = 0, sb = 0;
        const uchar4* row = tile + ty * tileW + lane;
        for (int k = 0; k <= 2 * radius; ++k) {
            const uchar4 t = row[k];
            sr += t.x; sg += t.y; sb += t.z;
        }
        hsum[ty * kBoxBlockX + lane] = make_uint2(sr | (sg << 16), sb);
    }
    __syncthreads();

    // 3. vertical sums + divide by the in-bounds tap count (reference "norm")
    const int gx = x0 + lane;
    if (gx >= W) return;
    const int cntx = min(gx + radius, W - 1) - max(gx - radius, 0) + 1;
#pragma unroll
    for (int i = 0; i < kTileY / 4; ++i) {
        const int oy = wave * (kTileY / 4) + i;
        const int gy = y0 + oy;
        if (gy >= H) break;
        unsigned sr = 0, sg = 0, sb = 0;
        for (int k = 0; k <= 2 * radius; ++k) {
            const uint2 h = hsum[(oy + k) * kBoxBlockX + lane];
            sr += h.x & 0xffffu; sg += h.x >> 16; sb += h.y;
        }
        const int cnty = min(gy + radius, H - 1) - max(gy - radius, 0) + 1;
        const float norm = (float)(cntx * cnty);
        uchar4 o;
        if (GRAY) {
            const unsigned char g = (unsigned char)((float)sr / norm);
            o = make_uchar4(g, g, g, 0);
        } else {
            o = make_uchar4((unsigned char)((float)sr / norm), (unsigned char)((float)sg / norm),
                            (unsigned char)((float)sb / norm), 0);
        }
        out[(size_t)gy * W + gx] = o;
    }
}
#endif

// ---- scan kernel: radii up to 32 at a cost that does not grow with the radius -----------------------------------------
// box_lds_kernel (experiments build) adds 2r+1 taps per row sum and 2r+1 row sums per pixel (r = 8 at 4K: ~250 VALU instructions and 51 LDS
// reads per pixel, 100-110 us).  Here
//   - a wave takes a staged row of 64 + 2r <= 128 pixels, two per lane, packs each pixel as (R | G << 16, B) and forms the
//     row's PREFIX sums with a DPP scan over the lanes' pair sums (row_shr 1/2/4/8, row_bcast 15/31: six v_add_u32_dpp per
//     word); a window's row sum is then prefix[x + 2r] - prefix[x - 1], two reads from the wave's LDS scratch row (fields
//     stay below 2^16: 128 pixels of 255);
//   - a thread walks down 8 output rows of its column with a RUNNING vertical sum: + the row sum that enters, - the one that
//     leaves (32-bit per channel: a window sum reaches (2r+1)^2 * 255);
//   - the quotient is floor((s + 1/2) / n) as one fma with v_rcp_f32(n): s and n are integers, so (s + 1/2) / n is at least
//     1/(2n) >= 1.18e-4 away from every integer and the two roundings (rcp 1 ulp, fma 1/2 ulp of a value <= 255) move it by
//     less than 5e-5; for integers floor((s + 1/2) / n) = floor(s / n), which is what the reference's
//     (unsigned char)((float)s / (float)n) gives (box_stream_kernel's comment).
// Same result as every other box kernel, bit for bit (tests/test_box_gpu.py: radii 5, 9, 24, 30 and the ragged shapes).
constexpr int kScanMaxRadius = 32;       // 32 output rows per workgroup, 8 per wave (64 rows: 46.9 us instead of 40.1 at radius 8, 4K --
                                         // 41 KB of LDS leave a CU three workgroups instead of six)

__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

template <bool GRAY, int kScanRows>
__global__ __launch_bounds__(256) void box_scan_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out, int W, int H, int radius)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char box_lds[];
    const int tileW = kBoxBlockX + 2 * radius, tileH = kScanRows + 2 * radius;
    uint2* hs = reinterpret_cast<uint2*>(box_lds);                      // [tileH][64] row sums of the 64 windows: (R | G << 16, B)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* P = hs + tileH * kBoxBlockX + wave * (tileW + 1);            // this wave's prefix row: P[i] = pixels 0 .. i-1
    const int x0 = blockIdx.x * kBoxBlockX, y0 = blockIdx.y * kScanRows;

    // A lane takes pixels 2 lane and 2 lane + 1 of the staged row (64 + 2r <= 128 of them); the rows of a wave are fetched four
    // at a time before any of them is scanned (a row at a time, the pass waited for one memory round trip per row).
    const int tx = 2 * lane, gxa = x0 - radius + tx, gxb = gxa + 1;
    const bool oka = tx < tileW && gxa >= 0 && gxa < W, okb = tx + 1 < tileW && gxb >= 0 && gxb < W;
    const unsigned* in32 = reinterpret_cast<const unsigned*>(in);
    for (int ty0 = wave; ty0 < tileH; ty0 += 16) {
        unsigned ua[4], ub[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = y0 - radius + ty0 + 4 * k;
            const bool rowok = ty0 + 4 * k < tileH && gy >= 0 && gy < H;
            ua[k] = (rowok && oka) ? in32[(size_t)gy * W + gxa] : 0u;               // out-of-frame cells add nothing
            ub[k] = (rowok && okb) ? in32[(size_t)gy * W + gxb] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ty = ty0 + 4 * k;
            if (ty >= tileH) break;                                                  // wave-uniform
            const unsigned a0 = (ua[k] & 0xffu) | ((ua[k] & 0xff00u) << 8), b0 = (ua[k] >> 16) & 0xffu;
            const unsigned a1 = a0 + ((ub[k] & 0xffu) | ((ub[k] & 0xff00u) << 8)), b1 = b0 + ((ub[k] >> 16) & 0xffu);
            const unsigned ea = wave_inclusive_scan(a1) - a1, eb = wave_inclusive_scan(b1) - b1;      // the lanes before this one
            if (tx < tileW)     P[tx + 1] = make_uint2(ea + a0, eb + b0);
            if (tx + 1 < tileW) P[tx + 2] = make_uint2(ea + a1, eb + b1);
            if (lane == 0) P[0] = make_uint2(0u, 0u);
            __builtin_amdgcn_wave_barrier();                            // (one wave's LDS accesses execute in order)
            const uint2 hi = P[lane + 2 * radius + 1], lo = P[lane];
            hs[ty * kBoxBlockX + lane] = make_uint2(hi.x - lo.x, hi.y - lo.y);
            __builtin_amdgcn_wave_barrier();                            // before the next row overwrites P
        }
    }
    __syncthreads();

    const int gx = x0 + lane;
    if (gx >= W) return;
    const int cntx = min(gx + radius, W - 1) - max(gx - radius, 0) + 1;
    const int oy0 = wave * (kScanRows / 4);
    unsigned sr = 0u, sg = 0u, sb = 0u;
    for (int k = 0; k <= 2 * radius; ++k) {
        const uint2 h = hs[(oy0 + k) * kBoxBlockX + lane];
        sr += h.x & 0xffffu; sg += h.x >> 16; sb += h.y;
    }
    for (int i = 0; i < kScanRows / 4; ++i) {
        const int oy = oy0 + i, gy = y0 + oy;
        if (gy >= H) break;
        const int cnty = min(gy + radius, H - 1) - max(gy - radius, 0) + 1;
        const float inv = __builtin_amdgcn_rcpf((float)(cntx * cnty)), half = 0.5f * inv;
        const unsigned qr = (unsigned)__builtin_fmaf((float)sr, inv, half);
        uchar4 o;
        if (GRAY) o = make_uchar4((unsigned char)qr, (unsigned char)qr, (unsigned char)qr, 0);
        else o = make_uchar4((unsigned char)qr, (unsigned char)(unsigned)__builtin_fmaf((float)sg, inv, half),
                             (unsigned char)(unsigned)__builtin_fmaf((float)sb, inv, half), 0);
        out[(size_t)gy * W + gx] = o;
        if (i + 1 < kScanRows / 4) {
            const uint2 hn = hs[(oy + 2 * radius + 1) * kBoxBlockX + lane], ho = hs[oy * kBoxBlockX + lane];
            sr += (hn.x & 0xffffu) - (ho.x & 0xffffu); sg += (hn.x >> 16) - (ho.x >> 16); sb += hn.y - ho.y;
        }
    }
}

// ---- run kernel: radii 33 .. 127, cost independent of the radius per output row (the reference takes any radius,
// src/filter.cu:34; (2r+1)^2 taps per pixel from global memory -- the direct kernel -- is 16 641 loads per pixel at r = 64) -----------
// A WAVE owns 64 columns x a band of rows and needs nobody else: no barriers.
//   - hsum(y): the 64 windows' row sums of row y.  The staged row is 64 + 2r <= 318 pixels, taken in chunks of 64 (one pixel per
//     lane); per chunk a DPP inclusive scan per channel (32-bit fields) + the carry of the chunks before it give the row's prefix
//     sums into the wave's LDS row; a window's sum is prefix[x + 2r + 1] - prefix[x].  Pixels outside the frame count 0.
//   - V(y) = sum of hsum over rows y - r .. y + r: built once at the top of the band (2r + 1 rows), then a RUNNING sum:
//     + hsum(y + r + 1) - hsum(y - r).  The leaving row is scanned again rather than kept (a ring of 2r + 1 row sums would be
//     up to 200 KB per workgroup); both rows' chunks are fetched before either is scanned.
//   - quotient: the reference's own (unsigned char)((float)s / (float)n) -- s < 2^24 for r <= 127, so the integer sum is exactly
//     what its float accumulation holds, in any order.
// Bit-exact like every other box kernel (tests/test_box_gpu.py: radii 33, 40, 64, 127).
constexpr int kRunMaxChunks = 5;         // ceil((64 + 2 * 127) / 64)

__device__ __forceinline__ void run_row_prefix(const unsigned* __restrict__ in32, const int W, const int H, const int gy, const int xs,
                                               const int nchunks, const int lane, unsigned (&px)[kRunMaxChunks])
{
    const bool rowok = gy >= 0 && gy < H;                 // wave-uniform
#pragma unroll
    for (int c = 0; c < kRunMaxChunks; ++c) {
        const int gx = xs + c * 64 + lane;
        px[c] = (c < nchunks && rowok && gx >= 0 && gx < W) ? in32[(size_t)gy * W + gx] : 0u;
    }
}

// prefix sums of one staged row into P[0 .. nchunks*64] (3 channels); then this lane's window sums
template <bool GRAY>
__device__ __forceinline__ void run_row_sums(const unsigned (&px)[kRunMaxChunks], const int nchunks, const int lane, const int radius,
                                             uint4* P, unsigned& hr, unsigned& hg, unsigned& hb)
{
    unsigned cr = 0u, cg = 0u, cb = 0u;
    if (lane == 0) P[0] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int c = 0; c < kRunMaxChunks; ++c) {
        if (c >= nchunks) break;                          // wave-uniform
        const unsigned r = px[c] & 0xffu, g = (px[c] >> 8) & 0xffu, b = (px[c] >> 16) & 0xffu;
        const unsigned sr = wave_inclusive_scan(r) + cr;
        unsigned sg = 0u, sb = 0u;
        if (!GRAY) { sg = wave_inclusive_scan(g) + cg; sb = wave_inclusive_scan(b) + cb; }
        P[c * 64 + lane + 1] = make_uint4(sr, sg, sb, 0u);
        cr = (unsigned)__builtin_amdgcn_readlane((int)sr, 63);
        if (!GRAY) { cg = (unsigned)__builtin_amdgcn_readlane((int)sg, 63); cb = (unsigned)__builtin_amdgcn_readlane((int)sb, 63); }
    }
    __builtin_amdgcn_wave_barrier();                      // (one wave's LDS accesses execute in order)
    const uint4 hi = P[lane + 2 * radius + 1], lo = P[lane];
    hr = hi.x - lo.x; hg = hi.y - lo.y; hb = hi.z - lo.z;
    __builtin_amdgcn_wave_barrier();                      // before the next row overwrites P
}

template <bool GRAY>
__global__ __launch_bounds__(256) void box_run_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out, int W, int H, int radius, int band_rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char box_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = 64 + 2 * radius, nchunks = (L + 63) / 64;
    uint4* P = reinterpret_cast<uint4*>(box_lds) + (size_t)wave * (nchunks * 64 + 1);
    const int x0 = blockIdx.x * 64, xs = x0 - radius;
    const int yb = (blockIdx.y * 4 + wave) * band_rows;
    if (yb >= H) return;
    const int ye = min(H, yb + band_rows);
    const unsigned* in32 = reinterpret_cast<const unsigned*>(in);
    unsigned vr = 0u, vg = 0u, vb = 0u;
    unsigned pa[kRunMaxChunks], pb[kRunMaxChunks];
    for (int y = yb - radius; y <= yb + radius; ++y) {    // (rows outside the frame add 0 and cost one empty scan)
        if (y < 0 || y >= H) continue;
        unsigned hr, hg, hb;
        run_row_prefix(in32, W, H, y, xs, nchunks, lane, pa);
        run_row_sums<GRAY>(pa, nchunks, lane, radius, P, hr, hg, hb);
        vr += hr; vg += hg; vb += hb;
    }
    const int gx = x0 + lane;
    const int cntx = min(gx + radius, W - 1) - max(gx - radius, 0) + 1;
    for (int y = yb; y < ye; ++y) {
        if (gx < W) {
            const int cnty = min(y + radius, H - 1) - max(y - radius, 0) + 1;
            const float n = (float)(cntx * cnty);
            uchar4 o;
            const unsigned char qr = (unsigned char)((float)vr / n);
            if (GRAY) o = make_uchar4(qr, qr, qr, 0);
            else o = make_uchar4(qr, (unsigned char)((float)vg / n), (unsigned char)((float)vb / n), 0);
            out[(size_t)y * W + gx] = o;
        }
        if (y + 1 < ye) {
            const int yin = y + radius + 1, yout = y - radius;
            run_row_prefix(in32, W, H, yin, xs, nchunks, lane, pa);
            run_row_prefix(in32, W, H, yout, xs, nchunks, lane, pb);
            unsigned hr, hg, hb;
            if (yin < H) { run_row_sums<GRAY>(pa, nchunks, lane, radius, P, hr, hg, hb); vr += hr; vg += hg; vb += hb; }
            if (yout >= 0) { run_row_sums<GRAY>(pb, nchunks, lane, radius, P, hr, hg, hb); vr -= hr; vg -= hg; vb -= hb; }
        }
    }
}

static int box_scan_rows(int) { return 32; }
static size_t box_scan_lds_bytes(int radius)
{
    const size_t tileW = kBoxBlockX + 2 * radius, tileH = box_scan_rows(radius) + 2 * radius;
    return (tileH * kBoxBlockX + 4 * (tileW + 1)) * sizeof(uint2);
}

// ---- stream kernel: the fast path for radius 1..4 (the reference runs radius 2) ---------------
// No LDS, no barriers.  A wave owns a 256-pixel column strip of one band of rows (a lane = 4
// consecutive pixels = one 16-byte load, 1 KB per wave and row) and walks down it:
//   - the pixels left / right of a lane's four come from the neighbouring lanes through DPP
//     wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1); lanes 0 and 63 load the strip's
//     4-pixel halo themselves and feed it in as the shift's `old` operand;
//   - channels are carried as packed 16-bit sums, (R | B<<16) and G: a horizontal sum is
//     <= 9*255 and a window sum <= 81*255 = 20655, so halves never carry into each other;
//   - the vertical sum is a running one over a ring of the last 2R+1 row sums held in registers
//     (the row loop is unrolled by the ring length, all indices static);
//   - rows are prefetched one ring length ahead (5 x 1 KB per wave in flight at radius 2);
//   - interior pixels divide by the constant (2R+1)^2 with one v_mul_hi_u32 (Granlund-Montgomery
//     magic number, exact for every sum < 2^16: checked on the host before the launch); pixels
//     whose window leaves the frame use the reference's float division.  Both equal
//     (unsigned char)((float)sum / (float)count): the quotient of two integers < 2^24 whose
//     fractional part is >= 1/count away from the next integer cannot be rounded across it.
// Algorithmic traffic 8 B/px; the pass is HBM-bound (66 MB at 4K).
__device__ __forceinline__ unsigned dpp_from_left(unsigned old, unsigned src)      // lane i <- lane i-1, lane 0 keeps old
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned dpp_from_right(unsigned old, unsigned src)     // lane i <- lane i+1, lane 63 keeps old
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x130, 0xf, 0xf, false);
}

template <int R, bool GRAY>
__global__ __launch_bounds__(256) void box_stream_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                         int W, int H, int nstrips, int band_rows, unsigned magic)
{
    constexpr int K = 2 * R + 1;                 // ring length = window height
    constexpr int U = K == 3 ? 6 : K;            // unroll / prefetch depth, a multiple of K
    const int lane = threadIdx.x & 63;
    // the wave index is the same in every lane: say so, and strip / band / every row number and
    // row test below stays in SGPRs (s_cmp + s_cbranch instead of v_cmp + exec masking)
    const int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strip = gw % nstrips, band = gw / nstrips;
    const int yb = band * band_rows;
    if (yb >= H) return;
    const int ye = min(H, yb + band_rows);
    const int x0 = strip * 256, x = x0 + lane * 4;
    const bool xin = x < W;                      // W % 4 == 0: a lane's four pixels are all in or all out
    const int W4 = W >> 2;
    // halo pixels of the strip: lane 0 fetches [x0-4, x0), lane 63 fetches [x0+256, x0+260)
    const bool hact = (lane == 0 && x0 > 0) || (lane == 63 && x0 + 256 < W);
    // Loads are UNCONDITIONAL (addresses clamped into the plane, values masked to 0 afterwards): with
    // loads inside branches the compiler cannot count them and drains the whole prefetch queue
    // (s_waitcnt vmcnt(0)) at every trip of the row loop.
    const int xc4 = min(x >> 2, W4 - 1);
    const int hc4 = min(max(lane == 0 ? (x0 >> 2) - 1 : (x0 >> 2) + 64, 0), W4 - 1);
    const unsigned own_mask = xin ? 0xffffffffu : 0u, halo_mask = hact ? 0xffffffffu : 0u;
    const int yfirst = yb - R, ylast = ye + R;   // input rows [yfirst, ylast)
    uint4 pre[U], preh[U];
    auto fetch = [&](const int y, uint4& own, uint4& halo) {
        const size_t row = (size_t)min(max(y, 0), H - 1) * (size_t)W4;
        own = in[row + (size_t)xc4];
        halo = in[row + (size_t)hc4];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(yfirst + u, pre[u], preh[u]);

    unsigned ring_rb[K][4], ring_g[K][4], v_rb[4], v_g[4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ring_rb[k][i] = 0u; ring_g[k][i] = 0u; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v_rb[i] = 0u; v_g[i] = 0u; }

    const bool lane_interior = x - R >= 0 && x + 3 + R < W;
    for (int base = yfirst; base < ylast; base += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int y = base + u;               // (rows >= ylast of the last trip are computed and dropped)
            const uint4 own = pre[u], halo = preh[u];
            const unsigned row_mask = (y >= 0 && y < H) ? 0xffffffffu : 0u;      // wave-uniform
            const unsigned om = own_mask & row_mask, hm = halo_mask & row_mask;
            // ---- packed words of the 4 + 2R pixels this lane's sums touch: index j <-> pixel x - R + j
            unsigned rb[4 + 2 * R], g[4 + 2 * R];
            const unsigned o[4] = { own.x & om, own.y & om, own.z & om, own.w & om };
            const unsigned hh[4] = { halo.x & hm, halo.y & hm, halo.z & hm, halo.w & hm };
            unsigned orb[4], og[4], hrb[4], hg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                orb[i] = GRAY ? (o[i] & 0xffu) : (o[i] & 0x00ff00ffu);  og[i] = (o[i] >> 8) & 0xffu;
                hrb[i] = GRAY ? (hh[i] & 0xffu) : (hh[i] & 0x00ff00ffu); hg[i] = (hh[i] >> 8) & 0xffu;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { rb[R + i] = orb[i]; g[R + i] = og[i]; }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                // pixel x-1-j = the left neighbour's pixel 3-j; pixel x+4+j = the right neighbour's pixel j
                rb[R - 1 - j] = dpp_from_left(hrb[3 - j], orb[3 - j]);
                rb[R + 4 + j] = dpp_from_right(hrb[j], orb[j]);
                if (!GRAY) {
                    g[R - 1 - j] = dpp_from_left(hg[3 - j], og[3 - j]);
                    g[R + 4 + j] = dpp_from_right(hg[j], og[j]);
                }
            }
            // ---- horizontal sums of the four pixels (sliding), then the vertical running sum
            unsigned h_rb[4], h_g[4];
            h_rb[0] = rb[0]; h_g[0] = GRAY ? 0u : g[0];
#pragma unroll
            for (int j = 1; j < K; ++j) { h_rb[0] += rb[j]; if (!GRAY) h_g[0] += g[j]; }
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                h_rb[i] = h_rb[i - 1] + rb[i + K - 1] - rb[i - 1];
                h_g[i] = GRAY ? 0u : h_g[i - 1] + g[i + K - 1] - g[i - 1];
            }
            constexpr int slot_of_u[6] = { 0 % K, 1 % K, 2 % K, 3 % K, 4 % K, 5 % K };
            const int slot = U == K ? u : slot_of_u[u < 6 ? u : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v_rb[i] += h_rb[i] - ring_rb[slot][i]; ring_rb[slot][i] = h_rb[i];
                if (!GRAY) { v_g[i] += h_g[i] - ring_g[slot][i]; ring_g[slot][i] = h_g[i]; }
            }
            // ---- refill this row's prefetch slot IN PLACE, now that its old contents are consumed (issued
            // before they are, the load needs other registers and the compiler rotates the slots with
            // copies at the loop end, which also have to wait for every load in flight)
            fetch(y + U, pre[u], preh[u]);
            // ---- output row y - R
            const int yo = y - R;
            if (yo < yb || yo >= ye) continue;   // wave-uniform
            const bool row_interior = yo - R >= 0 && yo + R < H;
            unsigned res[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned sr = v_rb[i] & 0xffffu, sb = v_rb[i] >> 16, sg = v_g[i];
                const unsigned qr = __umulhi(sr, magic);
                if (GRAY) res[i] = qr * 0x010101u;
                else      res[i] = qr | (__umulhi(sg, magic) << 8) | (__umulhi(sb, magic) << 16);
            }
            if (!(row_interior && lane_interior)) {
                // a window that leaves the frame: count the in-bounds taps, divide as the reference does
                const int cnty = min(yo + R, H - 1) - max(yo - R, 0) + 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = x + i;
                    const int cntx = min(px + R, W - 1) - max(px - R, 0) + 1;
                    const float norm = (float)(cntx * cnty);
                    const unsigned sr = v_rb[i] & 0xffffu, sb = v_rb[i] >> 16, sg = v_g[i];
                    const unsigned qr = (unsigned)(unsigned char)((float)sr / norm);
                    if (GRAY) res[i] = qr * 0x010101u;
                    else      res[i] = qr | ((unsigned)(unsigned char)((float)sg / norm) << 8) | ((unsigned)(unsigned char)((float)sb / norm) << 16);
                }
            }
            if (xin) out[(size_t)yo * (size_t)W4 + (size_t)(x >> 2)] = make_uint4(res[0], res[1], res[2], res[3]);
        }
    }
}

// magic number M with floor(s / n) == __umulhi(s, M) for every s < 2^16 (Granlund & Montgomery);
// verified exhaustively for the sums that can occur, 0 = no such number found
static unsigned box_magic(int n)
{
    int L = 0;
    while ((1 << L) < n) ++L;
    if (L < 1) return 0u;
    const unsigned long long m = ((1ull << (16 + L)) / (unsigned)n + 1ull) << (16 - L);
    if (m >> 32) return 0u;
    for (unsigned s = 0; s <= 255u * (unsigned)n; ++s)
        if ((unsigned)(((unsigned long long)s * m) >> 32) != s / (unsigned)n) return 0u;
    return (unsigned)m;
}

template <bool GRAY>
static bool launch_box_stream(const uchar4* in, uchar4* out, int W, int H, int radius, hipStream_t stream)
{
    static const bool disabled = tuning_env("RMD_BOX_STREAM", 1) == 0;    // A/B knob (experiments build)
    if (disabled || radius < 1 || radius > 4 || (W & 3) || !aligned_to(in, 16) || !aligned_to(out, 16)) return false;
    const unsigned magic = box_magic((2 * radius + 1) * (2 * radius + 1));
    if (!magic) return false;
    const int nstrips = (W + 255) / 256;
    // Band height: the pass is latency-bound until the chip holds ~4 waves per SIMD, so aim for ~4096
    // waves (11 rows at 4K, 36 at 8K; measured at 4K: 6 rows 23.1 us, 11 rows 18.4, 16 rows 19.4) and
    // accept that a band re-reads the 2R halo rows its neighbours fetch too (served by L2 / Infinity Cache).
    static const int band_env = tuning_env("RMD_BOX_BAND", 0);                // tuning knob (experiments build)
    int band_rows = (int)(((long long)H * nstrips + 4095) / 4096);
    band_rows = band_rows < 8 ? 8 : (band_rows > 64 ? 64 : band_rows);
    // the row loop runs in whole trips of U rows: make band + 2R a multiple of U (11 rows at 4K, radius 2)
    const int trip = (2 * radius + 1) == 3 ? 6 : 2 * radius + 1;
    band_rows = (band_rows + 2 * radius + trip - 1) / trip * trip - 2 * radius;
    if (band_env > 0) band_rows = band_env;
    const int nbands = (H + band_rows - 1) / band_rows;
    const int waves = nstrips * nbands;
    const dim3 grid((waves + 3) / 4);
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    switch (radius) {
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(box_stream_kernel<1, GRAY>), grid, dim3(256), 0, stream, in4, out4, W, H, nstrips, band_rows, magic); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(box_stream_kernel<2, GRAY>), grid, dim3(256), 0, stream, in4, out4, W, H, nstrips, band_rows, magic); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(box_stream_kernel<3, GRAY>), grid, dim3(256), 0, stream, in4, out4, W, H, nstrips, band_rows, magic); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(box_stream_kernel<4, GRAY>), grid, dim3(256), 0, stream, in4, out4, W, H, nstrips, band_rows, magic); break;
    }
    return true;
}

static size_t box_lds_bytes(int radius)
{
    const size_t tileW = kBoxBlockX + 2 * radius, tileH = kTileY + 2 * radius;
    return ((tileW * tileH * sizeof(uchar4) + 15) & ~(size_t)15) + tileH * kBoxBlockX * sizeof(uint2);
}

static int validate(const rmd_gbuffer& f, const rmd_filter_params& p, const char* who)
{
    if (f.shape.x <= 0 || f.shape.y <= 0) return fail(RMD_E_SHAPE, "%s: shape %dx%d is not positive", who, f.shape.x, f.shape.y);
    if ((long long)f.shape.x * f.shape.y > 0x7fffffffLL) return fail(RMD_E_SHAPE, "%s: shape overflows int", who);
    if (!f.render || !f.denoised) return fail(RMD_E_NULL, "%s: render/denoised plane is NULL", who);
    if (p.depth < 1) return fail(RMD_E_PARAM, "%s: depth %d < 1", who, p.depth);
    if (p.radius < 0) return fail(RMD_E_PARAM, "%s: radius %d < 0", who, p.radius);
    if (p.type < RMD_FILTER_AVERAGE || p.type > RMD_FILTER_WAVELET) return fail(RMD_E_PARAM, "%s: unknown filter type %d", who, p.type);
    if (p.depth > 1 && (!f.buffer[0] || !f.buffer[1])) return fail(RMD_E_BUFFER, "%s: depth %d needs both buffer[] planes", who, p.depth);
    if (f.render == f.denoised) return fail(RMD_E_BUFFER, "%s: render and denoised alias", who);
    if (!aligned_to(f.render, 4) || !aligned_to(f.denoised, 4)) return fail(RMD_E_ALIGN, "%s: planes must be 4-byte aligned", who);
    return RMD_OK;
}

// One launch per level with the reference's plane routing (src/filter.cu:24-25).
template <bool GRAY>
static int run_levels(const rmd_gbuffer& f, const rmd_filter_params& p, hipStream_t stream)
{
    const int W = f.shape.x, H = f.shape.y;
    for (int level = 0; level < p.depth; ++level) {
        const uchar4* in = reinterpret_cast<const uchar4*>(level == 0 ? f.render : f.buffer[level % 2]);
        uchar4* out = reinterpret_cast<uchar4*>(level == p.depth - 1 ? f.denoised : f.buffer[(level + 1) % 2]);
        static const bool scan_off = tuning_env("RMD_BOX_SCAN", 1) == 0;         // A/B knob (experiments build)
        if (launch_box_stream<GRAY>(in, out, W, H, p.radius, stream)) {
            // radius 1..4 on 16-byte aligned planes of a width that is a multiple of 4: the stream kernel
        } else if (!scan_off && p.radius >= 1 && p.radius <= kScanMaxRadius) {
            // every other radius up to 32, whatever cacheInput says (the result does not depend on it): prefix sums + running sums
            const int rows = box_scan_rows(p.radius);
            dim3 grid((W + kBoxBlockX - 1) / kBoxBlockX, (H + rows - 1) / rows);
            if (rows == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(box_scan_kernel<GRAY, 64>), grid, dim3(256), box_scan_lds_bytes(p.radius), stream, in, out, W, H, p.radius);
            else            hipLaunchKernelGGL(HIP_KERNEL_NAME(box_scan_kernel<GRAY, 32>), grid, dim3(256), box_scan_lds_bytes(p.radius), stream, in, out, W, H, p.radius);
#ifdef RMD_EXPERIMENTS
        } else if (scan_off && p.cacheInput && p.radius <= kMaxExactRadius && box_lds_bytes(p.radius) <= 64 * 1024) {
            // (RMD_BOX_SCAN=0 only: the 64 x 16 tile + halo kernel the scan kernel replaced)
            dim3 grid((W + kBoxBlockX - 1) / kBoxBlockX, (H + kTileY - 1) / kTileY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(box_lds_kernel<GRAY>), grid, dim3(256), box_lds_bytes(p.radius), stream,
                               in, out, W, H, p.radius);
#endif
        } else if (p.radius > kScanMaxRadius && p.radius <= kMaxExactRadius) {
            // radii 33 .. 127: running sums, a wave per 64 columns x band.  Bands of ~4r rows (the 2r + 1 rows that start a
            // band's sum are its overhead), fewer when that would leave CUs idle.
            const int strips = (W + 63) / 64;
            int band = 4 * p.radius;
            while (band > 16 && (long long)strips * ((H + band - 1) / band) < 4LL * 4 * device_cus()) band /= 2;
            const int nchunks = (64 + 2 * p.radius + 63) / 64;
            dim3 grid(strips, ((H + band - 1) / band + 3) / 4);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(box_run_kernel<GRAY>), grid, dim3(256), 4 * (nchunks * 64 + 1) * sizeof(uint4), stream,
                               in, out, W, H, p.radius, band);
        } else {
            // radius 0, and radii >= 128: there the reference's float accumulation rounds (sums reach 2^24), so its tap ORDER is
            // part of the result and only the kernel that follows it reproduces it
            dim3 grid((W + kBoxBlockX - 1) / kBoxBlockX, (H + kBoxBlockY - 1) / kBoxBlockY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(box_direct_kernel<GRAY>), grid, dim3(256), 0, stream, in, out, W, H, p.radius);
        }
        RMD_LAUNCH_CHECK("box filter launch");
    }
    return RMD_OK;
}

}  // namespace rmd

using namespace rmd;

extern "C" {

int rmd_filter_baseline(rmd_gbuffer frame, rmd_filter_params params, void* stream)
{
    if (int e = validate(frame, params, "rmd_filter_baseline")) return e;
    // The reference baseline ignores params.type (only AVERAGE exists, src/filter.cu:41).
    return run_levels<true>(frame, params, as_stream(stream));
}

int rmd_filter_tiled(rmd_gbuffer frame, rmd_filter_params params, void* stream)
{
    if (int e = validate(frame, params, "rmd_filter_tiled")) return e;
    if (params.type != RMD_FILTER_AVERAGE)     // GAUSSIAN / CROSS / WAVELET: csrc/weighted_filter.hip
        return run_weighted_levels(frame, params, as_stream(stream));
    // cacheInput selects nothing: every kernel stages what it needs, and both settings give the reference's cacheInput = false
    // result (its cached path is broken)
    return run_levels<false>(frame, params, as_stream(stream));
}

}  // extern "C"
