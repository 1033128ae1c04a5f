// Microbenchmark: can the waves of a CU share the lines of their epipolar bands in L1?  (tools/, not product)
//
// Models the gathers of pm_step: a wave owns a strip of 64 adjacent pixels and walks rows; per row and
// source it issues ONE 64-lane dword gather whose lanes land at random places t in [0, DISP) along a
// slanted epipolar segment of the source map (2-byte texels, pitch PITCH bytes, slope SLOPE rows per
// texel): byte address = base + floor(SLOPE * t) * PITCH + 2 * t.  Horizontally adjacent strips (OUTW
// pixels apart) have segments shifted by OUTW texels along the same line; the next row's segment is
// one map row lower.
//   mode 0: every wave an unrelated strip (random place per wave): no sharing possible
//   mode 1: the WGW waves of a workgroup are horizontally adjacent strips on the same rows, free running
//   mode 2: ... re-aligned with a barrier every SYNC rows
//   mode 3: ... and a barrier before EVERY source's gather (all waves of the workgroup issue source s
//           together: the working set of a phase is one source's lines)
// Output: ns per wave-level gather per CU (= the CU's L1 time per gather when that is the bound).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/l1_share tools/l1_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int PITCH = 2 * (1920 + 4);
constexpr int ROWS_MAP = 1084;
constexpr int DISP = 600;
constexpr int OUTW = 58;
constexpr int NSRC = 4;

__device__ __forceinline__ uint32_t fmix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

template <int WGW, int MODE, int SYNC>
__global__ __launch_bounds__(64 * WGW) void k(const char *__restrict__ maps, size_t map_bytes, int rows, int disp,
                                               float slope, uint32_t *out)
{
    extern __shared__ char pad_[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wg = blockIdx.x;
    // where this strip starts: a random column / row of the map, leaving room for the segment and the walk
    const uint32_t hs = fmix((MODE == 0 ? wg * WGW + wv : wg) * 2654435761u + 12345u);
    const int strips_x = (1920 - disp - WGW * OUTW) / OUTW;
    int x0 = (int)(hs % (uint32_t)strips_x) * OUTW;
    int y0 = (int)((hs >> 12) % (uint32_t)(ROWS_MAP - rows - 64 - (int)(slope * (disp + WGW * OUTW))));
    if (MODE != 0) x0 += wv * OUTW;                       // adjacent strips: same line, OUTW texels further
    uint32_t acc = 0;
    uint32_t h = fmix(wg * 977u + wv * 131u + lane * 7919u + 1u);
    for (int r = 0; r < rows; ++r) {
        if (MODE >= 2 && SYNC > 0 && r % SYNC == 0) __syncthreads();
        uint32_t w[NSRC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            if (MODE == 3) __syncthreads();
            h = h * 1664525u + 1013904223u;
            const int t = (int)(fmix(h) % (uint32_t)disp);           // this lane's place along the segment
            const int tx = x0 + lane + t;                            // (each lane's segment starts at its own column)
            const int ty = y0 + r + (int)(slope * (float)(tx - x0));
            const size_t off = (size_t)s * map_bytes + (size_t)ty * PITCH + 2u * (size_t)tx;
            __builtin_memcpy(&w[s], maps + off, 4);
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) acc ^= w[s];
    }
    out[(wg * WGW + wv) * 64 + lane] = acc + (uint32_t)pad_[0];
}

static char *g_maps;
static uint32_t *g_out;
static size_t g_map_bytes;

template <int WGW, int MODE, int SYNC>
void run(const char *name, int wgs_per_cu, int disp, float slope)
{
    const int ncu = 256, rows = 24, gens = 6;
    const int blocks = ncu * wgs_per_cu * gens;
    const size_t lds = 160 * 1024 / wgs_per_cu - 1024;       // residency cap: wgs_per_cu workgroups per CU
    hipFuncSetAttribute((const void *)k<WGW, MODE, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WGW, MODE, SYNC>), dim3(blocks), dim3(64 * WGW), lds, 0, g_maps, g_map_bytes, rows, disp, slope, g_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WGW, MODE, SYNC>), dim3(blocks), dim3(64 * WGW), lds, 0, g_maps, g_map_bytes, rows, disp, slope, g_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double gathers_per_cu = (double)wgs_per_cu * gens * WGW * rows * NSRC;
    const double ns = ms * 1e6 / gathers_per_cu;
    printf("%-58s %2d waves/CU disp %3d slope %.3f: %7.1f ns per gather per CU = %6.1f cyc   (%.3f ms)\n", name,
           wgs_per_cu * WGW, disp, slope, ns, ns * 2.1, ms);
    fflush(stdout);
}

int main()
{
    g_map_bytes = (size_t)PITCH * ROWS_MAP;
    hipMalloc(&g_maps, g_map_bytes * NSRC + 4096);
    hipMemset(g_maps, 1, g_map_bytes * NSRC + 4096);
    hipMalloc(&g_out, (size_t)256 * 16 * 8 * 64 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const int disp = pass == 0 ? 600 : 40;             // scattered hypotheses / converged ones
        const float slope = 0.073f;
        run<4, 0, 0>("unrelated strips, 4-wave workgroups", 4, disp, slope);
        run<4, 1, 0>("4 adjacent strips per workgroup, free running", 4, disp, slope);
        run<4, 2, 8>("4 adjacent strips, barrier every 8 rows", 4, disp, slope);
        run<4, 2, 1>("4 adjacent strips, barrier every row", 4, disp, slope);
        run<4, 3, 1>("4 adjacent strips, barrier before every gather", 4, disp, slope);
        run<4, 2, 1>("4 adjacent strips, barrier every row, 2 WGs/CU", 2, disp, slope);
        run<4, 2, 1>("4 adjacent strips, barrier every row, 1 WG/CU", 1, disp, slope);
        run<4, 3, 1>("4 adjacent strips, barrier before every gather, 1 WG/CU", 1, disp, slope);
        run<8, 2, 1>("8 adjacent strips, barrier every row", 2, disp, slope);
        run<8, 3, 1>("8 adjacent strips, barrier before every gather", 2, disp, slope);
        run<8, 3, 1>("8 adjacent strips, barrier before every gather, 1 WG/CU", 1, disp, slope);
        run<16, 1, 0>("16 adjacent strips, free running", 1, disp, slope);
        run<16, 2, 8>("16 adjacent strips, barrier every 8 rows", 1, disp, slope);
        run<16, 2, 1>("16 adjacent strips, barrier every row", 1, disp, slope);
        run<16, 3, 1>("16 adjacent strips, barrier before every gather", 1, disp, slope);
    }
    // flat epipolar lines (rectified-like pairs): how much is the slant?
    run<4, 2, 8>("4 adjacent strips, barrier every 8 rows", 4, 600, 0.0f);
    run<16, 3, 1>("16 adjacent strips, barrier before every gather", 1, 600, 0.0f);
    return 0;
}
