// Microbenchmark (tools/, not product code): does a cache-policy modifier change what a scattered
// gather costs on gfx950?  Same access pattern as gather_rate.hip's "dword, 64 random places"
// (every lane its own 128-byte line), the load issued as inline assembly with the modifier.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_policy tools/gather_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t fmix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

#define LOAD(POL)                                                                                  \
    asm volatile("global_load_dword %0, %1, off " POL : "=&v"(v[u]) : "v"(tab + r) : "memory")

template <int POLICY>
__global__ __launch_bounds__(64) void k(const char *__restrict__ tab, uint32_t mask, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t acc = 0;
    uint32_t h = fmix(wave * 2654435761u + 17u);
    for (int it = 0; it < iters; ++it) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            uint32_t r = fmix(h ^ (lane * 0x9E3779B9u)) & mask & ~3u;
            if constexpr (POLICY == 0) LOAD("");
            else if constexpr (POLICY == 1) LOAD("nt");
            else if constexpr (POLICY == 2) LOAD("sc0");
            else if constexpr (POLICY == 3) LOAD("sc1");
            else if constexpr (POLICY == 4) LOAD("sc0 sc1");
            else if constexpr (POLICY == 5) LOAD("sc0 sc1 nt");
            else if constexpr (POLICY == 6) LOAD("sc0 nt");
            else LOAD("sc1 nt");
        }
        // the results are tied through the wait, or the compiler consumes them before it
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                     :: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    out[wave * 64 + lane] = acc;
}

static char *g_tab;
static uint32_t *g_out;

template <int POLICY>
void run(const char *name, size_t table_bytes, int waves_per_cu)
{
    const int ncu = 256, iters = 400;
    const int blocks = ncu * waves_per_cu;
    const uint32_t mask = (uint32_t)(table_bytes - 1) & ~15u;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<POLICY>), dim3(blocks), dim3(64), 0, 0, g_tab, mask, 20, g_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<POLICY>), dim3(blocks), dim3(64), 0, 0, g_tab, mask, iters, g_out);
    hipEventRecord(e1);
    hipError_t err = hipEventSynchronize(e1);
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); fflush(stdout); return; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)waves_per_cu * iters * 8;
    const double ns = ms * 1e6 / loads_per_cu;
    printf("%-14s table %7.1f MB  %2d waves/CU: %7.1f ns per wave-load per CU = %6.1f cyc (%.2f cyc/line)\n",
           name, table_bytes / 1048576.0, waves_per_cu, ns, ns * 2.1, ns * 2.1 / 64);
    fflush(stdout);
}

int main()
{
    const size_t big = 2048ull << 20;
    printf("start\n"); fflush(stdout);
    if (hipMalloc(&g_tab, big) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(g_tab, 1, big);
    hipMalloc(&g_out, 256 * 32 * 64 * 4 * 4);
    for (size_t tb : {(size_t)2 << 20, (size_t)64 << 20, big}) {
        run<0>("default", tb, 16);
        run<1>("nt", tb, 16);
        run<2>("sc0", tb, 16);
        run<3>("sc1", tb, 16);
        run<4>("sc0 sc1", tb, 16);
        run<5>("sc0 sc1 nt", tb, 16);
        run<6>("sc0 nt", tb, 16);
        run<7>("sc1 nt", tb, 16);
    }
    return 0;
}
