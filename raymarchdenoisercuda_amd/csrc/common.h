// common.h — shared by every TU of librmd.so (host side of the C-ABI shim + device helpers).
// gfx950 only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include "../../include/rmd_api.h"

namespace rmd {

// ---- error plumbing -----------------------------------------------------------------------
// The reference checks no CUDA return code (include/vector.h:119-169, src/test.cu:73-89); here
// every entry point returns an int and records a message for rmd_last_error_string().
void set_error(const char* fmt, ...);
int  fail(int code, const char* fmt, ...);          // records message, returns code
int  hip_fail(hipError_t e, const char* what);      // records "<what>: <hipGetErrorString>"

#define RMD_HIP(call)                                                   \
    do {                                                                \
        hipError_t rmd_e_ = (call);                                     \
        if (rmd_e_ != hipSuccess) return ::rmd::hip_fail(rmd_e_, #call); \
    } while (0)

#define RMD_LAUNCH_CHECK(name)                                          \
    do {                                                                \
        hipError_t rmd_e_ = hipGetLastError();                          \
        if (rmd_e_ != hipSuccess) return ::rmd::hip_fail(rmd_e_, name); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kXcds = 8;         // MI355X accelerator complex dies (one L2 each)
constexpr int kCus  = 256;       // MI355X; device_cus() asks the current device
constexpr int kMaxDevices = 16;

// Facts and one-time setup that belong to a DEVICE, not to the process (a host program may drive
// several GPUs from one process through rmd_set_device).
int current_device();            // hipGetDevice, -1 on error
int device_cus();                // multiProcessorCount of the current device (cached)
// true exactly once per (device, key): guards per-device one-time calls such as hipFuncSetAttribute
bool first_use_on_device(const void* key);

// Kernel-tuning knobs (A/B switches of the measurement tools) are read from the environment ONLY in the
// experiments build (-DRMD_EXPERIMENTS); the product library always runs its defaults.
#ifdef RMD_EXPERIMENTS
int tuning_env(const char* name, int dflt);
#else
inline int tuning_env(const char*, int dflt) { return dflt; }
#endif

// Plane geometry shared by the SVGF kernels: planes hold global rows
// [buf_row0, buf_row0 + buf_rows) of a W x H frame.
struct Geom {
    int W, H, buf_row0, buf_rows;
};

int check_frame_geometry(const rmd_svgf_frame_desc* f);
// rows [lo,hi) clamped to the frame must be inside the buffer
int check_rows_in_buffer(const rmd_svgf_frame_desc* f, int lo, int hi, const char* what);

// Pass launchers behind rmd_svgf_temporal / rmd_svgf_variance.  rmd_svgf_frame fuses the V pass's
// pass-through copy into T: T writes its colour to t_color AND v_color, V then only rewrites the
// pixels on the spatial path (saves 32 B/px of the 64 B/px V would move).
// With f->v_tile_flags set, T also marks the 64x4 tiles (global tiling) that hold short-history
// pixels and V returns at once from every unmarked tile.
// FilterParams::type GAUSSIAN / CROSS / WAVELET on the uchar4 planes (csrc/weighted_filter.hip)
int run_weighted_levels(const rmd_gbuffer& f, const rmd_filter_params& p, hipStream_t stream);
// sparse_t_color: T writes t_color only inside the tiles it flags and V takes every other pixel of its windows from
// v_color.  Decided ONCE per frame by the caller (variance_reads_sparse_t_color) and handed to both launchers;
// launch_variance refuses it (RMD_E_PARAM) unless it runs the tile kernel, the only form that honours it.
int launch_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream, bool fused,
                    bool sparse_t_color);
int launch_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, void* stream, bool fused,
                    bool sparse_t_color);
// T + V of a whole frame in one launch (no statistics, var_radius 3): a workgroup that finds short-history pixels in its tile
// recomputes T on the tile's 3-pixel halo and runs V for them itself (svgf_temporal.hip)
// g8 != NULL: the 8-bit front end of rmd_svgf_gbuffer_frame (pixel_convert.h) -- inputs from the GBuffer's uchar4 planes, f->nd written
struct GBuffer8;
int launch_temporal_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1, int v_row0, int v_row1, void* stream,
                             const GBuffer8* g8 = nullptr);
bool variance_reads_sparse_t_color(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, bool fused);
// float4 illumination x uchar4 albedo -> uchar4 (convert_synth.hip; rmd_convert_f32_to_u8's arithmetic on the GBuffer's own albedo plane)
int launch_modulate_to_u8(const float* in, const void* albedo8, void* out8, size_t pixels, void* stream);

}  // namespace rmd

// ---- device helpers -------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace rmd {

__device__ __forceinline__ size_t pix_index(const Geom& g, int x, int y)
{
    return (size_t)(y - g.buf_row0) * (size_t)g.W + (size_t)x;
}

__device__ __forceinline__ float lum3(float r, float g, float b)
{
    // same operation order as the oracle (oracle/svgf_oracle.c lum3), no contraction
    return 0.2126f * r + 0.7152f * g + 0.0722f * b;
}

__device__ __forceinline__ bool is_zero3(float4 n) { return n.x == 0.0f && n.y == 0.0f && n.z == 0.0f; }

// v_exp_f32 / v_log_f32 (base 2, 1 ulp): the edge-stopping weights are evaluated in the log2
// domain, w = exp2(log2 k + sigma_n*log2(max(0,n.n)) - (w_z + w_l)*log2 e)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kNegInf = -__builtin_huge_valf();

// wave64 butterfly sum (DPP/ds_bpermute under the hood); every lane ends with the total
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace rmd
#endif
