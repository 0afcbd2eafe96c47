// valu_issue.hip — VALU issue rate of one gfx950 SIMD measured IN the kernel.
//
// Every workgroup declares 96 KiB of LDS so exactly one fits on a CU, and has 256*W threads so
// every SIMD of that CU holds exactly W waves.  Each wave brackets a long straight-line block
// of ONE instruction kind (8 independent destinations, 128 instructions per loop trip) with
// s_memtime (shader-clock ticks) and s_memrealtime (100 MHz), so the result needs no assumed
// clock and no assumption on where workgroups land:
//     cycles per instruction per SIMD = median over CUs of (last end - first start) / (N * W)
//     clock                           = dt_cycles / dt_real * 100 MHz
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o build/valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X)  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16x8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { FMA, MUL, ADD, PKFMA, PKMUL, PKADD, EXP, LOG, RCP, FMA_MOD, FMA_LIT, FMA_SGPR, MIX_TAP, FMA_EXP_1_4, PKFMA_OPSEL, MAX3, MED3, FMA_DPP, CVT, SUB, MAXF, MINF, FMAC, MOV, CNDMASK, FMA_CLAMP, MUL_CLAMP, FMA_INL, ADDU, LSHL, ANDB, MADU24, FMAMK, MUL_SGPR, FMA_2DIFF, DOT4U8, DOT2I16, CVTUB, PKSUBI16, PERM, MADI24, SADU8, N_MODES };
static const char* kNames[N_MODES] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
    "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_fma_f32 -|a|", "v_fma_f32 literal", "v_fma_f32 sgpr", "mix 12 valu + exp + log",
    "4 fma : 1 exp", "v_pk_fma_f32 op_sel", "v_max3_f32", "v_med3_f32", "v_add_f32 dpp row_shr", "v_cvt_f32_i32", "v_sub_f32", "v_max_f32", "v_min_f32", "v_fmac_f32", "v_mov_b32", "v_cndmask_b32", "v_fma_f32 clamp", "v_mul_f32 clamp(vop3)", "v_fma_f32 inline 1.0", "v_add_u32", "v_lshlrev_b32", "v_and_b32", "v_mad_u32_u24", "v_fmamk_f32", "v_mul_f32 sgpr", "v_fma_f32 3 distinct srcs", "v_dot4_u32_u8", "v_dot2_i32_i16", "v_cvt_f32_ubyte1", "v_pk_sub_i16", "v_perm_b32", "v_mad_i32_i24", "v_sad_u8" };

constexpr int kPerTrip = 128;

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* stamps, float* sink, int trips, float sarg)
{
    extern __shared__ unsigned char lds[];
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i + 1.0f; p[i] = f2{ a[i], a[i] + 0.5f }; }
    const float m = 1.0001f, c = 0.0001f;
    const f2 m2 = { m, m }, c2 = { c, c };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < trips; ++it) {
#define I_FMA(i)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_MUL(i)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define I_ADD(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#define I_EXP(i)   asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define I_LOG(i)   asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define I_RCP(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define I_FMAMOD(i) asm volatile("v_fma_f32 %0, -|%0|, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_FMALIT(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd" : "+v"(a[i]) : "v"(m));
#define I_FMASGPR(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sarg), "v"(c));
#define I_PKOPSEL(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(m2), "v"(c2));
#define I_MAX3(i)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_MED3(i)  asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_DPP(i)   asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
#define I_CVT(i)   asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
#define I_SUB(i)   asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MAXF(i)  asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MINF(i)  asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define I_FMAC(i)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
#define I_CND(i)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
#define I_FMACL(i) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_MULCL(i) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(m));
#define I_FMAINL(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(m));
#define I_ADDU(i)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_LSHL(i)  asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define I_ANDB(i)  asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_MADU(i)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1" : "+v"(a[i]) : "v"(c));
#define I_MULS(i)  asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sarg));
#define I_DOT4(i)  asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_DOT2(i)  asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
#define I_PKSUB(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define I_PERM(i)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_MADI(i)  asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_SAD(i)   asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define I_FMA3(i)  asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(m), "v"(c));
        if (MODE == FMA)   { REP16x8(I_FMA) }
        if (MODE == MUL)   { REP16x8(I_MUL) }
        if (MODE == ADD)   { REP16x8(I_ADD) }
        if (MODE == PKFMA) { REP16x8(I_PKFMA) }
        if (MODE == PKMUL) { REP16x8(I_PKMUL) }
        if (MODE == PKADD) { REP16x8(I_PKADD) }
        if (MODE == EXP)   { REP16x8(I_EXP) }
        if (MODE == LOG)   { REP16x8(I_LOG) }
        if (MODE == RCP)   { REP16x8(I_RCP) }
        if (MODE == FMA_MOD)  { REP16x8(I_FMAMOD) }
        if (MODE == FMA_LIT)  { REP16x8(I_FMALIT) }
        if (MODE == FMA_SGPR) { REP16x8(I_FMASGPR) }
        if (MODE == PKFMA_OPSEL) { REP16x8(I_PKOPSEL) }
        if (MODE == MAX3)  { REP16x8(I_MAX3) }
        if (MODE == MED3)  { REP16x8(I_MED3) }
        if (MODE == FMA_DPP) { REP16x8(I_DPP) }
        if (MODE == CVT)   { REP16x8(I_CVT) }
        if (MODE == SUB)   { REP16x8(I_SUB) }
        if (MODE == MAXF)  { REP16x8(I_MAXF) }
        if (MODE == MINF)  { REP16x8(I_MINF) }
        if (MODE == FMAC)  { REP16x8(I_FMAC) }
        if (MODE == MOV)   { REP16x8(I_MOV) }
        if (MODE == CNDMASK) { REP16x8(I_CND) }
        if (MODE == FMA_CLAMP) { REP16x8(I_FMACL) }
        if (MODE == MUL_CLAMP) { REP16x8(I_MULCL) }
        if (MODE == FMA_INL) { REP16x8(I_FMAINL) }
        if (MODE == ADDU)  { REP16x8(I_ADDU) }
        if (MODE == LSHL)  { REP16x8(I_LSHL) }
        if (MODE == ANDB)  { REP16x8(I_ANDB) }
        if (MODE == MADU24) { REP16x8(I_MADU) }
        if (MODE == FMAMK) { REP16x8(I_FMAMK) }
        if (MODE == MUL_SGPR) { REP16x8(I_MULS) }
        if (MODE == FMA_2DIFF) { REP16x8(I_FMA3) }
        if (MODE == DOT4U8)  { REP16x8(I_DOT4) }
        if (MODE == DOT2I16) { REP16x8(I_DOT2) }
        if (MODE == CVTUB)   { REP16x8(I_CVTUB) }
        if (MODE == PKSUBI16) { REP16x8(I_PKSUB) }
        if (MODE == PERM)    { REP16x8(I_PERM) }
        if (MODE == MADI24)  { REP16x8(I_MADI) }
        if (MODE == SADU8)   { REP16x8(I_SAD) }
        if (MODE == MIX_TAP) {
            // the shape of one a-trous tap for a pixel pair: 8 x (12 plain + 1 exp + 1 log + 2 pk) = 128
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                I_FMA(0) I_FMA(1) I_FMA(2) I_LOG(3) I_FMA(4) I_FMA(5) I_FMA(6) I_EXP(7)
                I_PKFMA(0) I_FMA(0) I_FMA(1) I_FMA(2) I_PKFMA(1) I_FMA(4) I_FMA(5) I_FMA(6)
            }
        }
        if (MODE == FMA_EXP_1_4) {
#pragma unroll
            for (int q = 0; q < 25; ++q) { I_FMA(0) I_FMA(1) I_FMA(2) I_FMA(4) I_EXP(7) }
            I_FMA(0) I_FMA(1) I_FMA(2)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) sink[threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[3 * wave] = t0;
        stamps[3 * wave + 1] = t1;
        stamps[3 * wave + 2] = r1 - r0;
    }
}

template <int MODE>
void run(unsigned long long* d_st, float* d_sink)
{
    const int trips = 400;
    printf("%-26s", kNames[MODE]);
    for (int wps : { 1, 2, 3, 4 }) {
        const int threads = 256 * wps, wgs = 256;
        const int waves = wgs * threads / 64;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        for (int rep = 0; rep < 3; ++rep)   // the last repetition is the one read back (clocks have settled)
            hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 96 * 1024, 0, d_st, d_sink, trips, 1.0001f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(3 * waves);
        hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        // per CU (= per workgroup): span from the first wave's start to the last wave's end
        const int wpw = threads / 64;
        std::vector<double> cyc(wgs), clk(waves), first(wgs);
        for (int g = 0; g < wgs; ++g) {
            unsigned long long lo = ~0ull, hi = 0, dmin = ~0ull;
            for (int w = g * wpw; w < (g + 1) * wpw; ++w) {
                lo = std::min(lo, st[3 * w]); hi = std::max(hi, st[3 * w + 1]);
                dmin = std::min(dmin, st[3 * w + 1] - st[3 * w]);
            }
            cyc[g] = (double)(hi - lo) / ((double)trips * kPerTrip * wps);
            first[g] = (double)dmin / ((double)trips * kPerTrip);
        }
        for (int w = 0; w < waves; ++w) clk[w] = (double)(st[3 * w + 1] - st[3 * w]) / (double)st[3 * w + 2] * 0.1;   // GHz
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        std::sort(first.begin(), first.end());
        printf("  W=%d: %5.2f (1st wave %5.2f/inst) @%4.2f GHz", wps, cyc[wgs / 2], first[wgs / 2], clk[waves / 2]);
    }
    printf("\n");
}

int main()
{
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, 3 * 8 * 256 * 16 * 8);
    hipMalloc(&d_sink, 4096 * 4);
    printf("cycles per wave64 instruction per SIMD = (last wave end - first wave start of a CU) / (instructions per wave x W waves per SIMD), median over the 256 CUs\n");
    run<FMA>(d_st, d_sink);
    run<MUL>(d_st, d_sink);
    run<ADD>(d_st, d_sink);
    run<FMA_MOD>(d_st, d_sink);
    run<FMA_LIT>(d_st, d_sink);
    run<FMA_SGPR>(d_st, d_sink);
    run<MAX3>(d_st, d_sink);
    run<MED3>(d_st, d_sink);
    run<FMA_DPP>(d_st, d_sink);
    run<CVT>(d_st, d_sink);
    run<PKFMA>(d_st, d_sink);
    run<PKFMA_OPSEL>(d_st, d_sink);
    run<PKMUL>(d_st, d_sink);
    run<PKADD>(d_st, d_sink);
    run<EXP>(d_st, d_sink);
    run<LOG>(d_st, d_sink);
    run<RCP>(d_st, d_sink);
    run<FMA_EXP_1_4>(d_st, d_sink);
    run<MIX_TAP>(d_st, d_sink);
    run<SUB>(d_st, d_sink);
    run<MAXF>(d_st, d_sink);
    run<MINF>(d_st, d_sink);
    run<FMAC>(d_st, d_sink);
    run<FMAMK>(d_st, d_sink);
    run<FMA_2DIFF>(d_st, d_sink);
    run<FMA_CLAMP>(d_st, d_sink);
    run<MUL_CLAMP>(d_st, d_sink);
    run<FMA_INL>(d_st, d_sink);
    run<MUL_SGPR>(d_st, d_sink);
    run<MOV>(d_st, d_sink);
    run<CNDMASK>(d_st, d_sink);
    run<ADDU>(d_st, d_sink);
    run<LSHL>(d_st, d_sink);
    run<ANDB>(d_st, d_sink);
    run<MADU24>(d_st, d_sink);
    run<DOT4U8>(d_st, d_sink);
    run<DOT2I16>(d_st, d_sink);
    run<CVTUB>(d_st, d_sink);
    run<PKSUBI16>(d_st, d_sink);
    run<PERM>(d_st, d_sink);
    run<MADI24>(d_st, d_sink);
    run<SADU8>(d_st, d_sink);
    return 0;
}
