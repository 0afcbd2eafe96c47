// valu_rate.hip — how fast does one SIMD of gfx950 issue f32 VALU work, by instruction kind and
// by waves per SIMD?  Decides whether the a-trous inner loop should be written with packed
// (v_pk_*) or scalar f32 ops and how much occupancy it needs.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{ a[i], a[i] + 0.5f }; }
    const float m = 1.0001f, c = 0.0001f;
    const float2v m2 = { m, m }, c2 = { c, c };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_fma_f32 %0, -|%0|, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 8) {   // dependent chain: 1 accumulator
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
            }
            if (MODE == 9) {   // 2 independent chains
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    for (int wps : { 1, 2, 3, 4, 8 }) {
        dim3 grid(256 * wps);
        hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = wps * iters * 8; cycles at 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("  w/SIMD=%d: %5.2f cyc/inst", wps, cyc / ((double)wps * iters * 8));
    }
    printf("\n");
}

int main()
{
    float* d;
    hipMalloc(&d, 4096);
    run<0>("v_fma_f32 x8 indep", d);
    run<1>("v_pk_fma_f32 x8 indep", d);
    run<4>("v_mul_f32 x8", d);
    run<5>("v_pk_mul_f32 x8", d);
    run<7>("v_fma_f32 -|a| mods", d);
    run<2>("v_exp_f32 x8", d);
    run<3>("v_log_f32 x8", d);
    run<6>("v_rcp_f32 x8", d);
    run<8>("v_fma_f32 1 dep chain", d);
    run<9>("v_fma_f32 2 chains", d);
    return 0;
}
