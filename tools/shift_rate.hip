// Microbenchmark 2: cross-lane shift primitives and slow VALU ops on gfx950 (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float c, const float* lutg)
{
    __shared__ float lut[256];
    for (int j = 0; j < 4; ++j) lut[threadIdx.x * 4 + j] = lutg[threadIdx.x * 4 + j];
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.37f + i + 1.0f;
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x101, 0xf, 0xf, true)) + c;          // row_shl:1
            else if (KIND == 1) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute((lane + 3) << 2, __float_as_int(a[i]))) + c;           // ds_bpermute
            else if (KIND == 2) a[i] = lut[(__float_as_uint(a[i]) >> 3) & 0xFF] + a[i];                                                 // random LDS lut read
            else if (KIND == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + c;
            else if (KIND == 4) a[i] = __builtin_amdgcn_sqrtf(a[i]) + c;
            else if (KIND == 5) a[i] = c / a[i];                                                                                        // IEEE division expansion
            else if (KIND == 6) a[i] = __builtin_sqrtf(a[i] + c);                                                                       // IEEE sqrt expansion
            else if (KIND == 7) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x130, 0xf, 0xf, true));     // v_mov wave_shl
            else if (KIND == 8) a[i] = __shfl_down(a[i], 1) + c;
            else if (KIND == 9) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x130, 0xf, 0xf, true)) + c;     // wave_shl:1 + add
            else if (KIND == 10) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x111, 0xf, 0xf, true)) + c;    // row_shr:1 + add
            else if (KIND == 11) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x138, 0xf, 0xf, true)) + c;    // wave_shr:1 + add
            else if (KIND == 12) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x103, 0xf, 0xf, true)) + c;    // row_shl:3 + add
            else if (KIND == 13) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x121, 0xf, 0xf, true)) + c;    // row_ror:1 + add
            else if (KIND == 14) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x142, 0xa, 0xf, true)) + c;    // row_bcast:15 + add
            else if (KIND == 15) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x134, 0xf, 0xf, true)) + c;    // wave_rol:1 + add
            else if (KIND == 16) a[i] = a[i] + c;                                                                                         // plain add
            else if (KIND == 17) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xf, 0xf, true)) + c;     // quad_perm [1,0,3,2] + add
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name)
{
    const int iters = 4000, blocks = 256 * 4 * 8;
    float *out, *lutg; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&lutg, 1024); hipMemset(lutg, 0, 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, 0.5f, lutg);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.5f, lutg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s %.3f ns per (op+add) per SIMD at 8 waves/SIMD (plain add alone = ~1.0)\n", name, ms * 1e6 / ((double)iters * 16 * 8));
    hipFree(out); hipFree(lutg);
}
int main() {
    run<16>("plain add"); run<9>("dpp wave_shl:1 + add"); run<10>("dpp row_shr:1 + add"); run<11>("dpp wave_shr:1 + add");
    run<12>("dpp row_shl:3 + add"); run<13>("dpp row_ror:1 + add"); run<14>("dpp row_bcast:15 + add"); run<15>("dpp wave_rol:1 + add");
    run<17>("dpp quad_perm + add");
    run<0>("dpp row_shl:1 + add"); run<7>("v_mov dpp wave_shl:1"); run<1>("ds_bpermute + add"); run<8>("__shfl_down + add");
    run<2>("lds lut read + add"); run<3>("v_rcp_f32 + add"); run<4>("v_sqrt_f32 + add"); run<5>("IEEE divide"); run<6>("add + IEEE sqrt");
    return 0;
}
