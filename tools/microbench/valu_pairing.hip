// valu_pairing.hip — when do two waves of one gfx950 SIMD issue "fast" f32 ops side by side?
//
// tools/microbench/valu_issue.hip shows two classes of VALU ops: plain f32 fma/mul/add (a wave issues one
// per ~4.6 cycles, but two waves of a SIMD overlap them: ~2.3 cycles per op per SIMD) and everything else
// (v_pk_*, transcendental, SGPR operand ...: 4.2 / 8.2 cycles, never overlapped).  A real kernel mixes
// the classes; this probe runs instruction PATTERNS shaped like the a-trous tap loop on W waves per SIMD
// and reports SIMD cycles per pattern, next to the two bounds
//     paired   = 2.3 F + 4.2 P + 8.2 T        (every plain op finds a partner)
//     unpaired = 4.2 F + 4.2 P + 8.2 T
// One workgroup per CU (96 KiB LDS), 256*W threads, timing by s_memtime inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_pairing.hip -o build/valu_pairing
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define F(i)  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(i) & 7]) : "v"(m), "v"(c));
#define P(i)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i) & 7]) : "v"(m2), "v"(c2));
#define L(i)  asm volatile("v_log_f32 %0, %0" : "+v"(a[(i) & 7]));
#define E(i)  asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i) & 7]));

// patterns: every one is 3 P (cosine) + 2 L + 10 F (exponents) + 2 E + 6 P (accumulate) = 23 instructions,
// the work of one tap for a pixel pair, in different orders
#define PAT_KERNEL  P(0) P(1) P(2) L(0) L(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) E(4) E(5) P(3) P(4) P(5) P(6) P(7) P(0)
// two taps interleaved: plain runs of 20
#define PAT_RUN2    P(0) P(1) P(2) P(3) P(4) P(5) L(0) L(1) L(2) L(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) \
                    E(0) E(1) E(2) E(3) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3)
// plain ops spread between the others
#define PAT_SPREAD  P(0) F(2) P(1) F(3) P(2) F(4) L(0) F(5) L(1) F(6) E(4) F(7) E(5) F(0) P(3) F(1) P(4) F(2) P(5) F(3) P(6) P(7) P(0)
#define PAT_ALLF    F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6)
#define PAT_ALLP    P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6)
#define PAT_ALLT    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(0) L(1) L(2) L(3) L(4) L(5) L(6)
#define X4(p) p p p p

enum { M_KERNEL, M_RUN2, M_SPREAD, M_F_VS_P, M_F_VS_T, M_F_VS_F, M_KERNEL_PRIO, M_RUN2_PRIO, N_MODES };
static const char* kNames[N_MODES] = { "tap order of the kernel", "two taps, runs of 20 plain", "plain ops spread out",
    "half the waves all-plain, half all-pk", "half all-plain, half all-trans", "all waves all-plain", "kernel order, waves 4+ at prio 1",
    "runs of 20, waves 4+ at prio 1" };

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* stamps, float* sink, int trips)
{
    extern __shared__ unsigned char lds[];
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i + 1.0f; p[i] = f2{ a[i], a[i] + 0.5f }; }
    const float m = 1.0001f, c = 0.0001f;
    const f2 m2 = { m, m }, c2 = { c, c };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = wave >= 4 && ((wave >> 2) & 1);      // waves 4-7, 12-15: the second wave of each SIMD pair
    if ((MODE == M_KERNEL_PRIO || MODE == M_RUN2_PRIO) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
        if (MODE == M_KERNEL || MODE == M_KERNEL_PRIO) { X4(PAT_KERNEL) }
        if (MODE == M_RUN2 || MODE == M_RUN2_PRIO)     { PAT_RUN2 PAT_RUN2 }
        if (MODE == M_SPREAD) { X4(PAT_SPREAD) }
        if (MODE == M_F_VS_P) { if (second) { X4(PAT_ALLP) } else { X4(PAT_ALLF) } }
        if (MODE == M_F_VS_T) { if (second) { X4(PAT_ALLT) } else { X4(PAT_ALLF) } }
        if (MODE == M_F_VS_F) { X4(PAT_ALLF) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) sink[threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t0;
        stamps[2 * w + 1] = t1;
    }
}

template <int MODE>
void run(unsigned long long* d_st, float* d_sink)
{
    const int trips = 300;
    printf("%-40s", kNames[MODE]);
    for (int wps : { 1, 2, 3, 4 }) {
        const int threads = 256 * wps, wgs = 256, wpw = threads / 64;
        const int waves = wgs * wpw;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 96 * 1024, 0, d_st, d_sink, trips);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(2 * waves);
        hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        // SIMD cycles per pattern (= one tap of a pixel pair, 23 instructions) = CU span / (4 patterns x trips x W)
        std::vector<double> cyc(wgs), fst(wgs), lst(wgs);
        for (int g = 0; g < wgs; ++g) {
            unsigned long long lo = ~0ull, hi = 0, dmin = ~0ull, dmax = 0;
            for (int w = g * wpw; w < (g + 1) * wpw; ++w) {
                lo = std::min(lo, st[2 * w]); hi = std::max(hi, st[2 * w + 1]);
                dmin = std::min(dmin, st[2 * w + 1] - st[2 * w]); dmax = std::max(dmax, st[2 * w + 1] - st[2 * w]);
            }
            cyc[g] = (double)(hi - lo) / (4.0 * trips * wps);
            fst[g] = (double)dmin / (4.0 * trips);
            lst[g] = (double)dmax / (4.0 * trips);
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(fst.begin(), fst.end()); std::sort(lst.begin(), lst.end());
        printf("  W=%d: %6.1f (waves %5.1f..%5.1f)", wps, cyc[wgs / 2], fst[wgs / 2], lst[wgs / 2]);
    }
    printf("\n");
}

int main()
{
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, 2 * 8 * 256 * 16);
    hipMalloc(&d_sink, 4096 * 4);
    printf("SIMD cycles per 23-instruction pattern (3+6 v_pk_fma, 10 v_fma, 2 v_log, 2 v_exp); bounds: paired %.1f, unpaired %.1f\n",
           2.3 * 10 + 4.2 * 9 + 8.2 * 4, 4.2 * 10 + 4.2 * 9 + 8.2 * 4);
    printf("(waves a..b = pattern time of the fastest / slowest wave of a CU)\n");
    run<M_KERNEL>(d_st, d_sink);
    run<M_RUN2>(d_st, d_sink);
    run<M_SPREAD>(d_st, d_sink);
    run<M_KERNEL_PRIO>(d_st, d_sink);
    run<M_RUN2_PRIO>(d_st, d_sink);
    run<M_F_VS_F>(d_st, d_sink);
    run<M_F_VS_P>(d_st, d_sink);
    run<M_F_VS_T>(d_st, d_sink);
    return 0;
}
