// convert_synth.hip — 8-bit <-> float plane conversion and the synthetic G-buffer generator.
//
// Conversion closes the loop between the reference's 8-bit planes (GBuffer::render/normal/
// albedo/denoised, reference include/gbuffer.h:9-12) and the float4 SVGF planes (SURVEY §8f.1,
// §8f.4).  The generator is the scene of SURVEY §8(d): not in the reference, which ships one
// static Cornell frame and no motion vectors.  Per pixel it uses only +,-,*,compare on fp32 (this
// TU is built with -ffp-contract=off) so it reproduces oracle/synth_oracle.c bit for bit and all
// ranks generate identical pixels with no transfers.
#include "common.h"
#include "pixel_convert.h"
#include <cmath>
#include <type_traits>

namespace rmd {

// ------------------------------------------------------------------------------ conversion
// (the per-pixel arithmetic lives in pixel_convert.h: the fused 8-bit ends of rmd_svgf_gbuffer_frame call the same functions)
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uchar4* __restrict__ in, float4* __restrict__ out, size_t n,
                                                        int renorm, float w_value)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = float4_from_u8(in[i], renorm != 0, w_value);
}

// albedo as a float4 plane (rmd_convert_f32_to_u8) or as the uchar4 plane of the GBuffer (the unfused tail of
// rmd_svgf_gbuffer_frame when the last iteration is also the history iteration)
template <class AlbedoT>
__global__ __launch_bounds__(256) void f32_to_u8_kernel(const float4* __restrict__ in, const AlbedoT* __restrict__ albedo,
                                                        uchar4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 al = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (albedo) {
            if constexpr (std::is_same<AlbedoT, uchar4>::value) al = float4_from_u8(albedo[i], false, 0.0f);
            else al = albedo[i];
        }
        out[i] = u8_from_float4(in[i], albedo != nullptr, al);
    }
}

// Demodulation: SVGF filters ILLUMINATION = radiance / albedo, so texture detail is not
// blurred; the albedo is multiplied back in rmd_convert_f32_to_u8.  One IEEE division per channel (the
// oracle's operation), the denominator floored at eps so black albedo does not produce infinities.
// radiance and out may be the same plane (rmd_api.h allows in == out): no __restrict__ on those two
__global__ __launch_bounds__(256) void demodulate_kernel(const float4* radiance, const float4* __restrict__ albedo,
                                                         float4* out, size_t n, float eps)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = demodulated(radiance[i], albedo[i], eps);
}

// ------------------------------------------------------------------------------ synthetic scene
constexpr int kRegions = 64;
struct Region {
    float cx, cy, hw, hh;
    float nx, ny, nz;
    float z0, ax, ay;
    float ar, ag, ab;
    float shade;
    float ux, uy;
};
struct RegionTable { Region r[kRegions + 1]; };

__host__ __device__ inline uint32_t hash32(uint32_t seed, uint32_t frame, uint32_t idx, uint32_t ch)
{
    uint32_t h = seed * 0x9E3779B1u;
    h ^= (frame + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= idx * 0xC2B2AE3Du;
    h ^= (ch + 1u) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
__host__ __device__ inline float unit_float(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// Region constants are evaluated on the host once per call (divisions / sqrt stay off the device).
static void build_region_table(uint32_t seed, int W, int H, RegionTable& t)
{
    const float lx = 0.3f, ly = 0.5f, lz = 0.8f;
    const float linv = 1.0f / std::sqrt(lx * lx + ly * ly + lz * lz);
    for (int k = 0; k < kRegions; ++k) {
        float u[15];
        for (int j = 0; j < 15; ++j) u[j] = unit_float(hash32(seed, 0xFFFFFFFFu, (uint32_t)k, (uint32_t)j));
        Region& r = t.r[k];
        r.cx = -64.0f + u[0] * (float)(W + 192);
        r.cy = -64.0f + u[1] * (float)(H + 128);
        r.hw = (0.03f + 0.12f * u[2]) * (float)W;
        r.hh = (0.03f + 0.12f * u[3]) * (float)H;
        const float dx = 2.0f * u[4] - 1.0f, dy = 2.0f * u[5] - 1.0f, dz = 0.5f + u[6];
        const float inv = 1.0f / std::sqrt(dx * dx + dy * dy + dz * dz);
        r.nx = dx * inv; r.ny = dy * inv; r.nz = dz * inv;
        r.z0 = 5.0f + 80.0f * u[7];
        r.ax = (u[8] - 0.5f) * 0.02f;
        r.ay = (u[9] - 0.5f) * 0.02f;
        r.ar = 0.2f + 0.7f * u[10]; r.ag = 0.2f + 0.7f * u[11]; r.ab = 0.2f + 0.7f * u[12];
        const float ndl = (r.nx * lx + r.ny * ly + r.nz * lz) * linv;
        r.shade = 0.2f + 0.8f * (ndl > 0.0f ? ndl : 0.0f);
        if (k % 4 == 0) {
            r.ux = (std::floor(u[13] * 9.0f) - 4.0f) * 0.25f;
            r.uy = (std::floor(u[14] * 9.0f) - 4.0f) * 0.25f;
        } else { r.ux = 0.0f; r.uy = 0.0f; }
    }
    Region& b = t.r[kRegions];
    b.cx = 0.0f; b.cy = 0.0f; b.hw = 3.0e38f; b.hh = 3.0e38f;
    b.nx = 0.0f; b.ny = 0.0f; b.nz = 1.0f;
    b.z0 = 90.0f; b.ax = 0.0f; b.ay = 0.005f;
    b.ar = b.ag = b.ab = 0.5f;
    b.shade = 0.6f; b.ux = 0.0f; b.uy = 0.0f;
}

struct SynthArgs {
    int W, buf_row0, buf_rows;
    uint32_t seed; int frame;
    float pan_x, pan_y;
    const RegionTable* table;
    float4* color; float4* nd; float2* motion; float4* albedo;
};

__global__ __launch_bounds__(256) void synth_kernel(SynthArgs a)
{
    __shared__ RegionTable tab;
    {
        const float* src = reinterpret_cast<const float*>(a.table);
        float* dst = reinterpret_cast<float*>(&tab);
        for (int i = threadIdx.x; i < (int)(sizeof(RegionTable) / sizeof(float)); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || r >= a.buf_rows) return;
    const int y = a.buf_row0 + r;
    const size_t o = (size_t)r * a.W + x;
    const float ff = (float)a.frame;
    const float wx = (float)x + ff * a.pan_x, wy = (float)y + ff * a.pan_y;
    int k = kRegions;
    float lx = wx, ly = wy;
    for (int j = kRegions - 1; j >= 0; --j) {
        const Region& q = tab.r[j];
        const float tx = wx - (q.cx + ff * q.ux), ty = wy - (q.cy + ff * q.uy);
        if (fabsf(tx) <= q.hw && fabsf(ty) <= q.hh) { k = j; lx = tx; ly = ty; break; }
    }
    const Region g = tab.r[k];
    float z = g.z0 + g.ax * lx + g.ay * ly;
    if (z < 1.0f) z = 1.0f;
    if (z > 100.0f) z = 100.0f;
    const uint32_t idx = (uint32_t)y * (uint32_t)a.W + (uint32_t)x;
    const bool firefly = hash32(a.seed, (uint32_t)a.frame, idx, 3u) < 0x0CCCCCCDu;
    const float light[3] = { 1.0f, 0.95f, 0.9f };
    float c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float u = unit_float(hash32(a.seed, (uint32_t)a.frame, idx, (uint32_t)ch));
        const float v = (g.shade * light[ch]) * (1.0f + 0.5f * (u - 0.5f));
        c[ch] = firefly ? 8.0f : v;
    }
    a.color[o] = make_float4(c[0], c[1], c[2], 0.0f);
    a.nd[o] = make_float4(g.nx, g.ny, g.nz, z);
    a.motion[o] = make_float2(a.pan_x - g.ux, a.pan_y - g.uy);
    if (a.albedo) a.albedo[o] = make_float4(g.ar, g.ag, g.ab, 1.0f);
}

// one small device table per device, rebuilt only when (seed, W, H) changes
static RegionTable* g_dev_table[16] = {};
static uint32_t g_tab_seed[16];
static int g_tab_w[16], g_tab_h[16];

}  // namespace rmd

using namespace rmd;

// float4 illumination x the GBuffer's uchar4 albedo -> uchar4 (rows of a plane; the tail of rmd_svgf_gbuffer_frame when its
// last a-trous launch cannot store bytes itself)
int rmd::launch_modulate_to_u8(const float* in, const void* albedo8, void* out8, size_t pixels, void* stream)
{
    if (pixels == 0) return RMD_OK;
    const unsigned blocks = (unsigned)((pixels + 255) / 256 < 2048 ? (pixels + 255) / 256 : 2048);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(f32_to_u8_kernel<uchar4>), dim3(blocks), dim3(256), 0, as_stream(stream),
                       (const float4*)in, (const uchar4*)albedo8, (uchar4*)out8, pixels);
    RMD_LAUNCH_CHECK("f32_to_u8_kernel<uchar4>");
    return RMD_OK;
}

extern "C" {

int rmd_convert_u8_to_f32(const rmd_uchar4* in, float* out, size_t pixels, int renormalize_xyz, float w_value, void* stream)
{
    if (!in || !out) return fail(RMD_E_NULL, "rmd_convert_u8_to_f32: NULL plane");
    if (!aligned_to(out, 16) || !aligned_to(in, 4)) return fail(RMD_E_ALIGN, "rmd_convert_u8_to_f32: misaligned plane");
    if (pixels == 0) return RMD_OK;
    const unsigned blocks = (unsigned)((pixels + 255) / 256 < 2048 ? (pixels + 255) / 256 : 2048);
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       (const uchar4*)in, (float4*)out, pixels, renormalize_xyz, w_value);
    RMD_LAUNCH_CHECK("u8_to_f32_kernel");
    return RMD_OK;
}

int rmd_convert_f32_to_u8(const float* in, const float* albedo, rmd_uchar4* out, size_t pixels, void* stream)
{
    if (!in || !out) return fail(RMD_E_NULL, "rmd_convert_f32_to_u8: NULL plane");
    if (!aligned_to(in, 16) || !aligned_to(albedo, 16) || !aligned_to(out, 4)) return fail(RMD_E_ALIGN, "rmd_convert_f32_to_u8: misaligned plane");
    if (pixels == 0) return RMD_OK;
    const unsigned blocks = (unsigned)((pixels + 255) / 256 < 2048 ? (pixels + 255) / 256 : 2048);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(f32_to_u8_kernel<float4>), dim3(blocks), dim3(256), 0, as_stream(stream),
                       (const float4*)in, (const float4*)albedo, (uchar4*)out, pixels);
    RMD_LAUNCH_CHECK("f32_to_u8_kernel");
    return RMD_OK;
}

int rmd_demodulate(const float* radiance, const float* albedo, float* out, size_t pixels, float eps, void* stream)
{
    if (!radiance || !albedo || !out) return fail(RMD_E_NULL, "rmd_demodulate: NULL plane");
    if (!aligned_to(radiance, 16) || !aligned_to(albedo, 16) || !aligned_to(out, 16)) return fail(RMD_E_ALIGN, "rmd_demodulate: misaligned plane");
    if (!(eps > 0.0f)) return fail(RMD_E_PARAM, "rmd_demodulate: eps must be > 0");
    if (pixels == 0) return RMD_OK;
    const unsigned blocks = (unsigned)((pixels + 255) / 256 < 2048 ? (pixels + 255) / 256 : 2048);
    hipLaunchKernelGGL(demodulate_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                       (const float4*)radiance, (const float4*)albedo, (float4*)out, pixels, eps);
    RMD_LAUNCH_CHECK("demodulate_kernel");
    return RMD_OK;
}

int rmd_synth_gbuffer(const rmd_synth_desc* d, float* color, float* nd, float* motion, float* albedo, void* stream)
{
    if (!d || !color || !nd || !motion) return fail(RMD_E_NULL, "rmd_synth_gbuffer: NULL argument");
    if (d->width <= 0 || d->height <= 0 || d->buf_rows <= 0 || d->buf_row0 < 0 || d->buf_row0 + d->buf_rows > d->height)
        return fail(RMD_E_SHAPE, "rmd_synth_gbuffer: bad geometry");
    if (!aligned_to(color, 16) || !aligned_to(nd, 16) || !aligned_to(motion, 8) || !aligned_to(albedo, 16))
        return fail(RMD_E_ALIGN, "rmd_synth_gbuffer: misaligned plane");
    int dev = 0;
    RMD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail(RMD_E_PARAM, "rmd_synth_gbuffer: device index %d unsupported", dev);
    const bool fresh = !g_dev_table[dev];
    if (fresh) RMD_HIP(hipMalloc((void**)&g_dev_table[dev], sizeof(RegionTable)));
    if (fresh || g_tab_seed[dev] != d->seed || g_tab_w[dev] != d->width || g_tab_h[dev] != d->height) {
        RegionTable host;
        build_region_table(d->seed, d->width, d->height, host);
        RMD_HIP(hipDeviceSynchronize());   // no generator launch may still be reading the old table
        RMD_HIP(hipMemcpy(g_dev_table[dev], &host, sizeof(RegionTable), hipMemcpyHostToDevice));
        g_tab_seed[dev] = d->seed; g_tab_w[dev] = d->width; g_tab_h[dev] = d->height;
    }
    SynthArgs a;
    a.W = d->width; a.buf_row0 = d->buf_row0; a.buf_rows = d->buf_rows;
    a.seed = d->seed; a.frame = d->frame; a.pan_x = d->pan_x; a.pan_y = d->pan_y;
    a.table = g_dev_table[dev];
    a.color = (float4*)color; a.nd = (float4*)nd; a.motion = (float2*)motion; a.albedo = (float4*)albedo;
    dim3 grid((d->width + 63) / 64, (d->buf_rows + 3) / 4);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, as_stream(stream), a);
    RMD_LAUNCH_CHECK("synth_kernel");
    return RMD_OK;
}

}  // extern "C"
