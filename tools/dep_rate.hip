// Microbenchmark: VALU issue interval of ONE wave on gfx950 as a function of the instruction-level
// parallelism in its stream (ILP independent fma chains) and of the waves resident per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o dep_rate tools/dep_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(64) void k(float *out, int iters, float b, float c)
{
    float a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / ILP; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP> void run()
{
    const int iters = 20000;
    float *out;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 16;
        printf("ILP=%2d waves/SIMD=%d  ns per instr per WAVE=%.2f  per SIMD=%.2f\n", ILP, wps, ms * 1e6 / n, ms * 1e6 / (n * wps));
    }
    hipFree(out);
}
int main() { run<1>(); run<2>(); run<4>(); run<16>(); return 0; }
