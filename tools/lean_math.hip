// Exhaustive check of lean reciprocal / sqrt sequences against the IEEE results on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/lean_math tools/lean_math.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float lean_rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(r, e, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ float lean_rcp1(float x)      // one correction only
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ float lean_sqrt(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float lean_sqrt2(float x)     // one more residual correction
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

template <int WHICH>
__global__ void check(uint32_t exp_lo, uint32_t exp_hi, unsigned long long* bad, uint32_t* first_bad)
{
    // all mantissas x exponents [exp_lo, exp_hi] x both signs (sign only for rcp)
    const unsigned long long n = (unsigned long long)(exp_hi - exp_lo + 1) << 23;
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t bits = (uint32_t)(((i >> 23) + exp_lo) << 23) | (uint32_t)(i & 0x7FFFFF);
        for (int sgn = 0; sgn < (WHICH < 2 ? 2 : 1); ++sgn) {
            float x = __uint_as_float(bits | (sgn ? 0x80000000u : 0u));
            float want, got;
            if (WHICH == 0) { want = 1.0f / x; got = lean_rcp(x); }
            else if (WHICH == 1) { want = 1.0f / x; got = lean_rcp1(x); }
            else if (WHICH == 2) { want = __builtin_sqrtf(x); got = lean_sqrt(x); }
            else { want = __builtin_sqrtf(x); got = lean_sqrt2(x); }
            if (__float_as_uint(want) != __float_as_uint(got)) { ++local; atomicMin(first_bad, bits); }
        }
    }
    if (local) atomicAdd(bad, local);
}

template <int WHICH> void run(const char* name, uint32_t lo, uint32_t hi)
{
    unsigned long long* bad; uint32_t* fb;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4); hipMemset(bad, 0, 8); hipMemset(fb, 0xff, 4);
    hipLaunchKernelGGL(check<WHICH>, dim3(4096), dim3(256), 0, 0, lo, hi, bad, fb);
    unsigned long long h; uint32_t f; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
    printf("%-28s biased exponents [%u,%u]: %llu mismatches of %llu (first bad bits 0x%08x)\n", name, lo, hi, h,
           (unsigned long long)(hi - lo + 1) << (WHICH < 2 ? 24 : 23), f);
    hipFree(bad); hipFree(fb);
}
int main()
{
    run<0>("rcp + 2 corrections", 1, 254);
    run<0>("rcp + 2 corrections", 32, 222);
    run<1>("rcp + 1 correction", 32, 222);
    run<2>("rsq-based sqrt", 1, 254);
    run<2>("rsq-based sqrt", 32, 222);
    run<3>("rsq-based sqrt + 1 corr", 32, 222);
    return 0;
}
