// pixel_convert.h -- the per-pixel arithmetic of the 8-bit ends of the SVGF path, ONE definition
// each: the standalone conversion kernels (convert_synth.hip: rmd_convert_u8_to_f32, rmd_demodulate, rmd_convert_f32_to_u8)
// and the fused ends of rmd_svgf_gbuffer_frame (the T+V launch reads the reference's uchar4 planes, include/gbuffer.h:9-12,
// the last a-trous launch writes `denoised`) call the same functions, so the one-call frame gives the bytes of the
// eight-call chain.  Every TU is built with -ffp-contract=off: what is written here is what is executed.
#pragma once
#include "common.h"

namespace rmd {

// (float)byte / 255.0f, correctly rounded, without the division: q = v * RN(1/255) is off by at most one ulp, the residual
// v - 255 q is exact in one fma, and the corrected quotient equals the IEEE quotient for all 256 bytes (checked exhaustively:
// tests/test_gbuffer_frame.py on the host arithmetic, tests/test_gbuffer_frame_gpu.py on the device).  3 instructions
// instead of the ~11 of v_div_scale / v_rcp / fma refinement / v_div_fixup, nine times per pixel in the front end.
__device__ __forceinline__ float unit_from_u8(unsigned char b)
{
    constexpr float r = 1.0f / 255.0f;
    const float v = (float)b;
    const float q = v * r;
    const float e = __builtin_fmaf(-q, 255.0f, v);
    return __builtin_fmaf(e, r, q);
}

// rmd_convert_u8_to_f32: c/255 per channel, optional renormalisation of xyz (normals: unit length or exactly zero, the input
// contract of the SVGF passes), w = w_value, or the plane's own w/255 when w_value < 0
__device__ __forceinline__ float4 float4_from_u8(const uchar4 p, const bool renorm, const float w_value)
{
    float x = unit_from_u8(p.x), y = unit_from_u8(p.y), z = unit_from_u8(p.z);
    if (renorm) {
        const float l2 = x * x + y * y + z * z;
        if (l2 > 0.0f) { const float inv = 1.0f / sqrtf(l2); x *= inv; y *= inv; z *= inv; }
    }
    return make_float4(x, y, z, w_value < 0.0f ? unit_from_u8(p.w) : w_value);
}

// rmd_demodulate: illumination = radiance / max(albedo, eps) per channel (IEEE divisions), w passed through
__device__ __forceinline__ float4 demodulated(const float4 c, const float4 al, const float eps)
{
    return make_float4(c.x / fmaxf(al.x, eps), c.y / fmaxf(al.y, eps), c.z / fmaxf(al.z, eps), c.w);
}

// The three values t = c * 255 + 0.5 of a pixel -> bytes, alpha 255: (unsigned char)clamp(t, 0, 255) with NaN -> 0, as
// v_floor_f32 + v_cvt_pk_u8_f32 per channel (the conversion saturates to [0, 255], takes NaN to 0 and inserts the byte into
// its place; floor(t) == trunc(t) wherever the clamp leaves t alone) instead of compare / select / min / convert / shift / or.
__device__ __forceinline__ uchar4 bytes_from_scaled(const float tr, const float tg, const float tb)
{
    unsigned v = 0xff000000u;
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(tr), 0u, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(tg), 1u, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(tb), 2u, v);
    return make_uchar4(v & 0xffu, (v >> 8) & 0xffu, (v >> 16) & 0xffu, v >> 24);
}

// rmd_convert_f32_to_u8: (optionally x albedo) -> x 255 + 0.5 -> clamp to [0, 255] -> truncate; alpha 255
__device__ __forceinline__ uchar4 u8_from_float4(float4 c, const bool modulate, const float4 al)
{
    if (modulate) { c.x = c.x * al.x; c.y = c.y * al.y; c.z = c.z * al.z; }
    return bytes_from_scaled(c.x * 255.0f + 0.5f, c.y * 255.0f + 0.5f, c.z * 255.0f + 0.5f);
}

// The same for TWO pixels at once (the a-trous kernel's vertical pair, whose VALU is the bound: every instruction is an issue
// slot), albedo given as the GBuffer's packed bytes: float4_from_u8 + u8_from_float4 lane for lane -- the same operations in
// the same order, so the same bytes -- with the arithmetic as packed f32 on the (A, B) register pair: per channel 2 byte
// converts + 6 packed + 2 floors + 2 converts instead of 2 x 11.
typedef float pc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void u8_pair_modulated(const float4 cA, const float4 cB, const unsigned albA, const unsigned albB,
                                                  unsigned& outA, unsigned& outB)
{
    constexpr float r = 1.0f / 255.0f;
    const float ca[3] = { cA.x, cA.y, cA.z }, cb[3] = { cB.x, cB.y, cB.z };
    unsigned vA = 0xff000000u, vB = 0xff000000u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const pc_f2 v = { (float)((albA >> (8 * k)) & 0xffu), (float)((albB >> (8 * k)) & 0xffu) };      // v_cvt_f32_ubyteK
        const pc_f2 q = v * pc_f2{ r, r };                                                              // unit_from_u8
        const pc_f2 e = __builtin_elementwise_fma(-q, pc_f2{ 255.0f, 255.0f }, v);
        const pc_f2 al = __builtin_elementwise_fma(e, pc_f2{ r, r }, q);
        const pc_f2 m = pc_f2{ ca[k], cb[k] } * al;                                                     // u8_from_float4
        const pc_f2 t = m * pc_f2{ 255.0f, 255.0f } + pc_f2{ 0.5f, 0.5f };
        vA = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(t.x), (unsigned)k, vA);
        vB = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(t.y), (unsigned)k, vB);
    }
    outA = vA; outB = vB;
}

// The uchar4 planes of the reference's GBuffer as the inputs / the output of an SVGF frame (rmd_svgf_gbuffer_frame).
// Same buffer geometry as the float planes of the call.
struct GBuffer8 {
    const uchar4* render;     // radiance
    const uchar4* albedo;
    const uchar4* normal;     // xyz: normal, renormalised; w: linear depth in 1/255 (opaque alpha = depth 1)
    uchar4* denoised;
    float albedo_eps;
};

}  // namespace rmd
