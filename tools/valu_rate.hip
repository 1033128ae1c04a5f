// Microbenchmark: issue cost of VALU instruction kinds on gfx950 as a function of waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters, float b, float c)
{
    float a[16]; float2_t p[8]; unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 977u + i; }
    for (int i = 0; i < 8; ++i) { p[i].x = a[2*i]; p[i].y = a[2*i+1]; }
    float2_t pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], b, c);
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pb, pc);
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) u[i] = u[i] * 0x85EBCA6Bu + 1u;
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x130, 0xf, 0xf, true)) + c;
        } else if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = a[i] + c;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i]; for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, int per_instr_ops)
{
    const int iters = 20000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&cyc, 256 * 4 * 8 * 8);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, 100, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        double instr_per_wave = (double)iters * per_instr_ops;
        // s_memtime ticks at 100 MHz? report both wall-derived and tick-derived
        printf("%-14s waves/SIMD=%d  wall=%.3f ms  ticks/wave=%.0f  wall ns per instr per SIMD=%.3f (x2.4GHz = %.2f cyc)\n",
               name, wps, ms, avg, ms * 1e6 / (instr_per_wave * wps), ms * 1e6 / (instr_per_wave * wps) * 2.4);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_fma_f32", 16); run<1>("v_pk_fma_f32", 8); run<2>("v_mul_lo+add", 16); run<3>("v_add_dpp", 16); run<4>("v_add_f32", 16);
    return 0;
}
