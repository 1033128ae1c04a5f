// Microbenchmark: what does a scattered gather cost on gfx950?  (tools/, not product code)
// Every wave issues `iters` x 8 independent 64-lane loads whose addresses follow a pattern, with a
// table that stays L2-resident (2 MB), Infinity-Cache-resident (64 MB) or neither (2 GB).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_rate tools/gather_rate.hip
// Output: ns per wave-level load instruction per CU and the implied cycles per lane at 2.1 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t fmix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// GROUP consecutive lanes read consecutive elements (GROUP = 1: every lane its own random place;
// 64: one fully coalesced access per wave).  BAND > 0: all lanes of an instruction fall inside one
// random window of BAND bytes (an epipolar band: 2 rows x ~600 px of a 2-byte map = ~2.4 KB).
template <typename T, int GROUP, int BAND>
__global__ __launch_bounds__(64) void k(const char *__restrict__ tab, uint32_t mask, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    T acc{};
    uint32_t h = fmix(wave * 2654435761u + 17u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;                        // wave-uniform stream
            uint32_t r;
            if (BAND > 0) {
                const uint32_t base = fmix(h) & mask;              // window start (uniform)
                r = base + (fmix(h ^ (lane / GROUP * 0x9E3779B9u)) % (uint32_t)BAND);
            } else {
                r = fmix(h ^ (lane / GROUP * 0x9E3779B9u)) & mask;
            }
            r = (r & ~(uint32_t)(sizeof(T) * GROUP - 1)) + (lane % GROUP) * sizeof(T);
            r &= mask;
            T v;
            __builtin_memcpy(&v, tab + r, sizeof(T));
            if constexpr (sizeof(T) <= 4) acc = (T)(acc ^ v);
            else if constexpr (sizeof(T) == 8) { acc.x ^= v.x; acc.y ^= v.y; }
            else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        }
    }
    uint32_t o;
    if constexpr (sizeof(T) <= 4) o = (uint32_t)acc;
    else if constexpr (sizeof(T) == 8) o = acc.x ^ acc.y;
    else o = acc.x ^ acc.y ^ acc.z ^ acc.w;
    out[wave * 64 + lane] = o;
}

// Calibration of FETCH_SIZE for sparse reads: one dword per STRIDE bytes, lanes on consecutive
// strides (so the request rate is not the limit).  If stride 128 takes as long as stride 64 over the
// same buffer, an L2 miss moves the whole 128-byte line; if it takes half as long, only a 64-byte
// sector.
template <int STRIDE>
__global__ __launch_bounds__(256) void sweep(const char *__restrict__ tab, size_t bytes, uint32_t *out)
{
    const size_t n = bytes / STRIDE;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v;
        __builtin_memcpy(&v, tab + i * STRIDE, 4);
        acc ^= v;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int STRIDE>
void run_sweep(const char *tab, size_t bytes, uint32_t *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((sweep<STRIDE>), dim3(256 * 8), dim3(256), 0, 0, tab, bytes, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((sweep<STRIDE>), dim3(256 * 8), dim3(256), 0, 0, tab, bytes, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("sweep: one dword per %3d bytes over %.1f GB: %.3f ms  (%.2f TB/s if whole 128-B lines move, %.2f TB/s if 64-B sectors)\n",
           STRIDE, bytes / 1e9, ms, bytes / (ms * 1e-3) / 1e12, (STRIDE >= 128 ? bytes / 2 : bytes) / (ms * 1e-3) / 1e12);
}

static char *g_tab;
static uint32_t *g_out;

template <typename T, int GROUP, int BAND>
void run(const char *name, size_t table_bytes, int waves_per_cu)
{
    const int ncu = 256, iters = 400;
    const int blocks = ncu * waves_per_cu;
    const uint32_t mask = (uint32_t)(table_bytes - 1) & ~15u;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, GROUP, BAND>), dim3(blocks), dim3(64), 0, 0, g_tab, mask, 20, g_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, GROUP, BAND>), dim3(blocks), dim3(64), 0, 0, g_tab, mask, iters, g_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)waves_per_cu * iters * 8;
    const double ns = ms * 1e6 / loads_per_cu;
    printf("%-34s table %7.1f MB  %2d waves/CU: %7.1f ns per wave-load per CU = %6.1f cyc (%.2f cyc/lane)  %.2f TB/s useful\n",
           name, table_bytes / 1048576.0, waves_per_cu, ns, ns * 2.1, ns * 2.1 / 64,
           64.0 * sizeof(T) * ncu / ns / 1e3);
}

int main()
{
    const size_t big = 2048ull << 20;
    hipMalloc(&g_tab, big);
    hipMemset(g_tab, 1, big);
    hipMalloc(&g_out, 256 * 32 * 64 * 4 * 4);
    run_sweep<4>(g_tab, big, g_out);
    run_sweep<64>(g_tab, big, g_out);
    run_sweep<128>(g_tab, big, g_out);
    run_sweep<256>(g_tab, big, g_out);
    for (size_t tb : {(size_t)2 << 20, (size_t)64 << 20, big}) {
        run<uint32_t, 1, 0>("dword, 64 random places", tb, 16);
        run<uint32_t, 2, 0>("dword, 32 places x 2 lanes", tb, 16);
        run<uint32_t, 4, 0>("dword, 16 places x 4 lanes", tb, 16);
        run<uint32_t, 16, 0>("dword, 4 places x 16 lanes", tb, 16);
        run<uint32_t, 64, 0>("dword, coalesced", tb, 16);
        run<uint32_t, 1, 2400>("dword, random in a 2.4 KB band", tb, 16);
        run<uint32_t, 1, 16384>("dword, random in a 16 KB band", tb, 16);
        run<uint16_t, 1, 0>("ushort, 64 random places", tb, 16);
        run<uint2, 1, 0>("dwordx2, 64 random places", tb, 16);
        run<uint4, 1, 0>("dwordx4, 64 random places", tb, 16);
    }
    for (int w : {4, 8, 24, 32}) run<uint32_t, 1, 0>("dword, 64 random places", (size_t)2 << 20, w);
    for (int w : {4, 8, 24, 32}) run<uint32_t, 1, 2400>("dword, random in a 2.4 KB band", (size_t)64 << 20, w);
    return 0;
}
