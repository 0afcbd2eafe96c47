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
//   - LDS kernel (cacheInput=true, the job of reference cacheTile src/filter.cu:60-85): a
//     64x16 output tile + halo is staged once in LDS with ONE consistent stride (the reference
//     stages with a rounded stride and reads with the unrounded one), then the box
//     sum is done separably on integers: all partial sums are < 2^24, so integer sums converted
//     to float equal the reference's float accumulation exactly, for any order.
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
static int run_levels(const rmd_gbuffer& f, const rmd_filter_params& p, bool use_lds, hipStream_t stream)
{
    const int W = f.shape.x, H = f.shape.y;
    for (int level = 0; level < p.depth; ++level) {
        const uchar4* in = reinterpret_cast<const uchar4*>(level == 0 ? f.render : f.buffer[level % 2]);
        uchar4* out = reinterpret_cast<uchar4*>(level == p.depth - 1 ? f.denoised : f.buffer[(level + 1) % 2]);
        if (use_lds) {
            dim3 grid((W + kBoxBlockX - 1) / kBoxBlockX, (H + kTileY - 1) / kTileY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(box_lds_kernel<GRAY>), grid, dim3(256), box_lds_bytes(p.radius), stream,
                               in, out, W, H, p.radius);
        } else {
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
    return run_levels<true>(frame, params, /*use_lds=*/false, as_stream(stream));
}

int rmd_filter_tiled(rmd_gbuffer frame, rmd_filter_params params, void* stream)
{
    if (int e = validate(frame, params, "rmd_filter_tiled")) return e;
    if (params.type != RMD_FILTER_AVERAGE)     // GAUSSIAN / CROSS / WAVELET: csrc/weighted_filter.hip
        return run_weighted_levels(frame, params, as_stream(stream));
    const bool lds_ok = params.radius <= kMaxExactRadius && box_lds_bytes(params.radius) <= 64 * 1024;
    return run_levels<false>(frame, params, params.cacheInput && lds_ok, as_stream(stream));
}

}  // extern "C"
