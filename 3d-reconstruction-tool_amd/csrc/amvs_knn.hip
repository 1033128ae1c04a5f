// amvs_knn.hip -- k-nearest-neighbour mean distances of a point cloud on the device: the expensive
// part of the stereo path's statistical outlier removal (reference:
// DenseStereoReconstructor._filter_outliers, dense_stereo.py:439-473, which calls scikit-learn's
// NearestNeighbors(n_neighbors=k).kneighbors on the host and averages distances[:, 1:]).
//
// Same numbers bit for bit (checked against scikit-learn in tests/test_hip_parity.py):
//   * scikit-learn's KD-tree evaluates the squared distance as ((dx*dx) + (dy*dy)) + (dz*dz) in
//     float64 without fused operations and returns sqrt of the k smallest, ascending, the query
//     point itself (distance 0) first;
//   * np.mean over the remaining k-1 values is numpy's pairwise summation: eight accumulators over
//     blocks of eight, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail added in order.
// The values do not depend on how ties are broken, so any exact k-nearest search gives them.  The
// search here is a uniform grid: points are binned into cells of edge h (counting sort), a thread
// per point visits the cells of growing Chebyshev shells around its own cell and stops after shell
// r once its k-th smallest squared distance is <= (r*h)^2 (everything not yet visited is farther).
#include "amvs_kernels.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace amvs {

namespace {

#define KCHK(call)                                 \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) return e_;           \
    } while (0)

constexpr int KNN_KMAX = 32;
constexpr int KNN_GMAX = 256;       // cells per axis (dense table of at most 2^24 cells)

struct Grid {
    double lo[3];
    double inv_h, h;
    int g[3];
};

__device__ __forceinline__ int cell_of(const Grid &gr, double p, int axis)
{
    int c = (int)floor((p - gr.lo[axis]) * gr.inv_h);
    c = c < 0 ? 0 : c;
    return c >= gr.g[axis] ? gr.g[axis] - 1 : c;
}

__global__ __launch_bounds__(256) void knn_count_kernel(const double *__restrict__ pts, long long n, Grid gr,
                                                        int *__restrict__ cell, int *__restrict__ count)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int cx = cell_of(gr, pts[3 * i], 0), cy = cell_of(gr, pts[3 * i + 1], 1), cz = cell_of(gr, pts[3 * i + 2], 2);
        const int c = (cz * gr.g[1] + cy) * gr.g[0] + cx;
        cell[i] = c;
        atomicAdd(&count[c], 1);
    }
}

// counting sort: point i goes to slot start[cell] + (its arrival order inside the cell)
__global__ __launch_bounds__(256) void knn_place_kernel(const double *__restrict__ pts, long long n,
                                                        const int *__restrict__ cell, const int *__restrict__ start,
                                                        int *__restrict__ cursor, double *__restrict__ sorted,
                                                        int *__restrict__ origin)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = cell[i];
        const int slot = start[c] + atomicAdd(&cursor[c], 1);
        sorted[3 * (long long)slot] = pts[3 * i];
        sorted[3 * (long long)slot + 1] = pts[3 * i + 1];
        sorted[3 * (long long)slot + 2] = pts[3 * i + 2];
        origin[slot] = (int)i;
    }
}

// numpy's pairwise summation of n < 128 doubles (numpy/_core/src/umath/loops_utils.h.src)
template <int N>
__device__ __forceinline__ double numpy_pairwise_sum(const double (&a)[KNN_KMAX], int first)
{
    if (N < 8) {
        double res = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) res += a[first + i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[first + j];
    constexpr int BODY = N - (N % 8);
#pragma unroll
    for (int i = 8; i < BODY; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += a[first + i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = BODY; i < N; ++i) res += a[first + i];
    return res;
}

template <int K>
__global__ __launch_bounds__(128) void knn_query_kernel(const double *__restrict__ sorted, long long n, Grid gr,
                                                        const int *__restrict__ start,   // [cells + 1]
                                                        const int *__restrict__ origin,
                                                        double *__restrict__ mean_out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const double qx = sorted[3 * q], qy = sorted[3 * q + 1], qz = sorted[3 * q + 2];
    const int cx = cell_of(gr, qx, 0), cy = cell_of(gr, qy, 1), cz = cell_of(gr, qz, 2);

    // the K smallest squared distances seen so far (unsorted) and the largest of them
    double best[KNN_KMAX];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = __builtin_inf();
    double worst = __builtin_inf();

    auto visit_cell = [&](int x, int y, int z) {
        const int c = (z * gr.g[1] + y) * gr.g[0] + x;
        const int b = start[c], e = start[c + 1];
        for (int p = b; p < e; ++p) {
            const double dx = qx - sorted[3 * (long long)p], dy = qy - sorted[3 * (long long)p + 1],
                         dz = qz - sorted[3 * (long long)p + 2];
            const double d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
            if (d2 < worst) {
                // replace one entry equal to the current largest, then find the new largest
                bool done = false;
                double w = -1.0;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bool hit = !done & (best[j] == worst);
                    best[j] = hit ? d2 : best[j];
                    done |= hit;
                    w = best[j] > w ? best[j] : w;
                }
                worst = w;
            }
        }
    };

    const int rmax = max(max(gr.g[0], gr.g[1]), gr.g[2]);
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, gr.g[0] - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, gr.g[1] - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gr.g[2] - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool face = abs(z - cz) == r || abs(y - cy) == r;
                if (face) {
                    for (int x = x0; x <= x1; ++x) visit_cell(x, y, z);
                } else {                     // interior rows of the shell: only its two x faces
                    if (cx - r >= 0) visit_cell(cx - r, y, z);
                    if (r > 0 && cx + r < gr.g[0]) visit_cell(cx + r, y, z);
                }
            }
        // every unvisited point lies at least r*h away (slightly shrunk against rounding of the binning)
        const double reach = (double)r * gr.h * (1.0 - 1e-9);
        if (worst <= reach * reach) break;
    }

    // ascending order, sqrt, drop the first (the query itself), numpy-order mean of the other K-1
#pragma unroll
    for (int i = 1; i < K; ++i)
#pragma unroll
        for (int j = K - 1; j >= i; --j) {
            const double a = best[j - 1], b = best[j];
            best[j - 1] = a < b ? a : b;
            best[j] = a < b ? b : a;
        }
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = sqrt(best[j]);
    mean_out[origin[q]] = numpy_pairwise_sum<K - 1>(best, 1) / (double)(K - 1);
}

template <int K>
hipError_t launch_query(const double *sorted, long long n, const Grid &gr, const int *start, const int *origin,
                        double *mean_out, hipStream_t st)
{
    hipLaunchKernelGGL((knn_query_kernel<K>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, sorted, n, gr,
                       start, origin, mean_out);
    return hipGetLastError();
}

}  // namespace

bool knn_supported(int k) { return k == 8 || k == 10 || k == 16 || k == 20 || k == 32; }

// points: host [n][3] float64; mean_out: host [n].  Needs n >= k.
hipError_t knn_mean_distance(const double *points, long long n, int k, double *mean_out, hipStream_t st)
{
    // bounding box and cell edge: about 8 points per cell if the points filled the box, refined below
    // from the occupancy actually found (reconstructed clouds are surfaces, not volumes)
    Grid gr{};
    double hi[3];
    for (int a = 0; a < 3; ++a) { gr.lo[a] = points[a]; hi[a] = points[a]; }
    for (long long i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            gr.lo[a] = std::min(gr.lo[a], points[3 * i + a]);
            hi[a] = std::max(hi[a], points[3 * i + a]);
        }
    double ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) { ext[a] = hi[a] - gr.lo[a]; emax = std::max(emax, ext[a]); }
    if (!(emax > 0.0) || !std::isfinite(emax)) emax = 1.0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], emax * 1e-3);
    double h = std::cbrt(vol * 8.0 / (double)n);
    const double h_min = emax / (KNN_GMAX - 2);      // keeps every cell index below KNN_GMAX without clamping
    h = std::max(h, h_min);

    double *d_pts = nullptr, *d_sorted = nullptr, *d_mean = nullptr;
    int *d_cell = nullptr, *d_count = nullptr, *d_start = nullptr, *d_origin = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_pts, (void *)d_sorted, (void *)d_mean, (void *)d_cell, (void *)d_count,
                        (void *)d_start, (void *)d_origin, d_tmp})
            if (p) (void)hipFree(p);
    };
#define KCHK_C(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) { cleanup(); return e_; }                \
    } while (0)
    KCHK_C(hipMalloc(&d_pts, sizeof(double) * 3 * n));
    KCHK_C(hipMalloc(&d_sorted, sizeof(double) * 3 * n));
    KCHK_C(hipMalloc(&d_mean, sizeof(double) * n));
    KCHK_C(hipMalloc(&d_cell, sizeof(int) * n));
    KCHK_C(hipMalloc(&d_origin, sizeof(int) * n));
    KCHK_C(hipMemcpyAsync(d_pts, points, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));

    const int bx = (int)std::min<long long>((n + 255) / 256, 4096);
    long long cells = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        gr.h = h; gr.inv_h = 1.0 / h;
        for (int a = 0; a < 3; ++a) gr.g[a] = std::max(1, std::min(KNN_GMAX, (int)std::floor(ext[a] / h) + 1));
        cells = (long long)gr.g[0] * gr.g[1] * gr.g[2];
        if (d_count) { (void)hipFree(d_count); d_count = nullptr; }
        if (d_start) { (void)hipFree(d_start); d_start = nullptr; }
        KCHK_C(hipMalloc(&d_count, sizeof(int) * (cells + 1)));
        KCHK_C(hipMalloc(&d_start, sizeof(int) * (cells + 1)));
        KCHK_C(hipMemsetAsync(d_count, 0, sizeof(int) * (cells + 1), st));
        hipLaunchKernelGGL(knn_count_kernel, dim3(bx), dim3(256), 0, st, d_pts, n, gr, d_cell, d_count);
        KCHK_C(hipGetLastError());
        // occupancy of the occupied cells decides whether the edge fits the data
        size_t need = 0;
        KCHK_C(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_count, d_start, (int)(cells + 1), st));
        if (need > tmp_bytes) {
            if (d_tmp) (void)hipFree(d_tmp);
            d_tmp = nullptr;
            KCHK_C(hipMalloc(&d_tmp, need));
            tmp_bytes = need;
        }
        KCHK_C(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_count, d_start, (int)(cells + 1), st));
        std::vector<int> cnt(cells);
        KCHK_C(hipMemcpyAsync(cnt.data(), d_count, sizeof(int) * cells, hipMemcpyDeviceToHost, st));
        KCHK_C(hipStreamSynchronize(st));
        long long occupied = 0;
        for (long long c = 0; c < cells; ++c) occupied += cnt[c] > 0;
        const double per_cell = (double)n / (double)std::max<long long>(occupied, 1);
        const bool at_limit = gr.g[0] == KNN_GMAX || gr.g[1] == KNN_GMAX || gr.g[2] == KNN_GMAX;
        if (per_cell > 24.0 && !at_limit && h > h_min) { h = std::max(h * 0.5, h_min); continue; }
        if (per_cell < 3.0 && cells > 1) { h *= 2.0; continue; }
        break;
    }
    KCHK_C(hipMemsetAsync(d_count, 0, sizeof(int) * (cells + 1), st));     // reused as the placement cursor
    hipLaunchKernelGGL(knn_place_kernel, dim3(bx), dim3(256), 0, st, d_pts, n, d_cell, d_start, d_count, d_sorted,
                       d_origin);
    KCHK_C(hipGetLastError());
    hipError_t e = hipErrorInvalidValue;
    switch (k) {
    case 8: e = launch_query<8>(d_sorted, n, gr, d_start, d_origin, d_mean, st); break;
    case 10: e = launch_query<10>(d_sorted, n, gr, d_start, d_origin, d_mean, st); break;
    case 16: e = launch_query<16>(d_sorted, n, gr, d_start, d_origin, d_mean, st); break;
    case 20: e = launch_query<20>(d_sorted, n, gr, d_start, d_origin, d_mean, st); break;
    case 32: e = launch_query<32>(d_sorted, n, gr, d_start, d_origin, d_mean, st); break;
    default: break;
    }
    KCHK_C(e);
    KCHK_C(hipMemcpyAsync(mean_out, d_mean, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    KCHK_C(hipStreamSynchronize(st));
    cleanup();
    return hipSuccess;
#undef KCHK_C
}

}  // namespace amvs
