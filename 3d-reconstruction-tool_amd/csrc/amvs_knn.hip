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
// search here: points are binned into a uniform grid over the 1st-99th percentile box (counting
// sort, cell edge h fitted to the occupancy); a thread per point visits the cells of Chebyshev shells
// 0..1 around its own cell and is done after shell r once its k-th smallest squared distance is
// <= (r*h)^2 (everything not yet visited is farther); a query left pending gets a wave that scans the box of
// cells within R = 2 shells cooperatively, and if the rule still fails a block that scans the boxes R = 4, 8,
// 16, ... (knn_box_kernel) until it holds or the box is the whole grid.
#define AMVS_TU_ID 8
#include "amvs_check.h"
#include "amvs_kernels.h"
#include "amvs_pool.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
constexpr int KNN_SHELLS = 2;       // Chebyshev shells a query walks on one grid level
#ifndef AMVS_KNN_DEBUG
#define AMVS_KNN_DEBUG false
#endif

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
        const int c = AMVS_IDX((cz * gr.g[1] + cy) * gr.g[0] + cx, (long long)gr.g[0] * gr.g[1] * gr.g[2]);
        cell[i] = c;
        atomicAdd(&count[c], 1);
    }
}

// counting sort: point i goes to slot start[cell] + (its arrival order inside the cell)
__global__ __launch_bounds__(256) void knn_place_kernel(const double *__restrict__ pts, long long n,
                                                        const int *__restrict__ cell, const int *__restrict__ start,
                                                        int *__restrict__ cursor, double *__restrict__ sorted,
                                                        int *__restrict__ origin, long long cells)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = AMVS_IDX(cell[i], cells);
        const int slot = AMVS_IDX(start[c] + atomicAdd(&cursor[c], 1), n);
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

// One more candidate for a list of the K smallest values kept in ASCENDING order: a chain of min / max pairs carries
// the new value to its place and drops the largest -- 2 K double-precision instructions, no compares, no selects
// (the first version replaced the current largest and searched the new one: 6-7 instructions per entry; the lists
// are what the search kernels spend their time on: 6.4e8 VALU wave-instructions per 500 000 queries).
template <int K>
__device__ __forceinline__ void knn_insert(double (&best)[KNN_KMAX], double &worst, double x)
{
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double lo = fmin(best[j], x);
        x = fmax(best[j], x);
        best[j] = lo;
    }
    worst = best[K - 1];
}

// The cell walk: every query walks at most `max_shells` Chebyshev shells of cells around its own;
// those that cover their k-th distance write their mean and clear the flag, the others stay pending
// for knn_box_kernel.
template <int K>
__global__ __launch_bounds__(128) void knn_query_kernel(const double *__restrict__ sorted, long long n, Grid gr,
                                                        const int *__restrict__ start,   // [cells + 1]
                                                        const int *__restrict__ origin,
                                                        unsigned char *__restrict__ pending,   // by original index
                                                        int max_shells,
                                                        double *__restrict__ mean_out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int self = AMVS_IDX(origin[q], n);
    if (!pending[self]) return;
    const double qx = sorted[3 * q], qy = sorted[3 * q + 1], qz = sorted[3 * q + 2];
    const int cx = cell_of(gr, qx, 0), cy = cell_of(gr, qy, 1), cz = cell_of(gr, qz, 2);

    // the K smallest squared distances seen so far, ascending, and the largest of them
    double best[KNN_KMAX];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = __builtin_inf();
    double worst = __builtin_inf();

    // the cells x_first .. x_last of grid row (y, z): consecutive in `start`, so their points are ONE range of the
    // sorted array -- one pair of look-ups per row of a shell's face instead of one per cell (45 instead of 125 for
    // shells 0..2: the walk is a chain of dependent loads, not arithmetic)
    auto visit_row = [&](int x_first, int x_last, int y, int z) {
        const long long ncell = (long long)gr.g[0] * gr.g[1] * gr.g[2];
        (void)ncell;
        const int c0 = AMVS_IDX((z * gr.g[1] + y) * gr.g[0] + x_first, ncell);
        const int c1 = AMVS_IDX((z * gr.g[1] + y) * gr.g[0] + x_last, ncell);
        const int b = AMVS_IDX(start[c0], n + 1), e = AMVS_IDX(start[c1 + 1], n + 1);
        for (int p = b; p < e; ++p) {
            const double dx = qx - sorted[3 * (long long)p], dy = qy - sorted[3 * (long long)p + 1],
                         dz = qz - sorted[3 * (long long)p + 2];
            const double d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
            if (d2 < worst) knn_insert<K>(best, worst, d2);
        }
    };

    const int gmax = max(max(gr.g[0], gr.g[1]), gr.g[2]);
    const int rmax = min(max_shells, gmax);
    bool done = false;
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, gr.g[0] - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, gr.g[1] - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gr.g[2] - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool face = abs(z - cz) == r || abs(y - cy) == r;
                if (face) {
                    visit_row(x0, x1, y, z);
                } else {                     // interior rows of the shell: only its two x faces
                    if (cx - r >= 0) visit_row(cx - r, cx - r, y, z);
                    if (r > 0 && cx + r < gr.g[0]) visit_row(cx + r, cx + r, y, z);
                }
            }
        // every unvisited point lies at least r*h away (slightly shrunk against rounding of the binning)
        const double reach = (double)r * gr.h * (1.0 - 1e-9);
        if (worst <= reach * reach) { done = true; break; }
    }
    if (!done && rmax >= gmax) done = true;        // every cell was visited
    if (!done) return;                             // stays pending: knn_box_kernel takes it

    // (ascending already) sqrt, drop the first (the query itself), numpy-order mean of the other K-1
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = sqrt(best[j]);
    mean_out[self] = numpy_pairwise_sum<K - 1>(best, 1) / (double)(K - 1);
    pending[self] = 0;
}

// Queries the cell walk of knn_query_kernel left pending: one 256-thread block per query scans the
// BOX of cells within R Chebyshev shells of the query's cell on the same (fine) grid -- the cells of a
// grid row are contiguous in the sorted point array, so a row of the box is one coalesced range; waves
// take rows, lanes take points, each thread keeps the K smallest of its share -- then extracts the K
// smallest overall, one per round, with a (value, thread) arg-min reduction, and tests the stopping rule of the cell walk: every point
// outside the box is at least R h away, so the K-th smallest distance found is final once it is
// <= R h.  If not, R doubles; a box that covers the grid is a full scan.  A single thread walking a
// coarser grid took ~1.2 us per query and the coarser grids had to be re-binned (6-7 ms each on a
// 277 k-point cloud); this takes one launch.
// NW = waves per block: 4 for the doubling boxes; ONE for the first cooperative pass, the box R = 2 (r_last = 2: a
// query it does not settle stays pending) -- the cells a single thread's walk would visit as shells 0..2, but scanned
// by 64 lanes: on the stereo path's cloud 12 % of the queries need shell 2, many of them next to crowded cells, and a
// thread walking 45 rows of up to hundreds of points each held its whole wave for 5 ms.
template <int K, int NW>
__global__ __launch_bounds__(64 * NW) void knn_box_kernel(const double *__restrict__ pts, const double *__restrict__ sorted,
                                                          Grid gr, const int *__restrict__ start, int r_first, int r_last,
                                                          const int *__restrict__ queries, unsigned char *__restrict__ pending,
                                                          double *__restrict__ mean_out, long long n)
{
    const int self = AMVS_IDX(queries[blockIdx.x], n);
    const double qx = pts[3 * (long long)self], qy = pts[3 * (long long)self + 1], qz = pts[3 * (long long)self + 2];
    const int cx = cell_of(gr, qx, 0), cy = cell_of(gr, qy, 1), cz = cell_of(gr, qz, 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double red_v[NW];
    __shared__ int red_t[NW];
    __shared__ double out[KNN_KMAX];
    for (int R = r_first;; R *= 2) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, gr.g[0] - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, gr.g[1] - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, gr.g[2] - 1);
        const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == gr.g[0] - 1 && y1 == gr.g[1] - 1 && z1 == gr.g[2] - 1;
        double best[KNN_KMAX];
#pragma unroll
        for (int j = 0; j < K; ++j) best[j] = __builtin_inf();
        double worst = __builtin_inf();
        const int ny = y1 - y0 + 1, rows = ny * (z1 - z0 + 1);
        // Rows of the box in batches of 64 per wave: lane l looks up the point range of row (batch + l) -- the
        // look-ups of a batch are independent loads in flight together --, then the non-empty rows are scanned one
        // after the other, lanes taking points.  (The boxes of the queries that end up here are sparse: a row-by-row
        // walk spent its time waiting for one dependent pair of look-ups per row, 1 089 rows per wave at R = 16.)
        for (int row0 = wave * 64; row0 < rows; row0 += NW * 64) {
            const int row = row0 + lane;
            int b = 0, e = 0;
            if (row < rows) {
                const int z = z0 + row / ny, y = y0 + row % ny;
                const long long base = ((long long)z * gr.g[1] + y) * gr.g[0];
                const long long cells1 = (long long)gr.g[0] * gr.g[1] * gr.g[2] + 1;       // entries of `start`
                (void)cells1;
                b = AMVS_IDX(start[AMVS_IDX(base + x0, cells1)], n + 1);
                e = AMVS_IDX(start[AMVS_IDX(base + x1 + 1, cells1)], n + 1);
            }
            unsigned long long todo = __ballot(e > b);
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int rb = __shfl(b, src), re = __shfl(e, src);
                for (int p = rb + lane; p < re; p += 64) {
                    const double dx = qx - sorted[3 * (long long)p], dy = qy - sorted[3 * (long long)p + 1],
                                 dz = qz - sorted[3 * (long long)p + 2];
                    const double d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
                    if (d2 < worst) knn_insert<K>(best, worst, d2);
                }
            }
        }
        // (own lists are ascending) the K smallest of the block, one per round
        int head = 0;
        for (int t = 0; t < K; ++t) {
            double v = __builtin_inf();
#pragma unroll
            for (int j = 0; j < K; ++j) v = (j == head) ? best[j] : v;
            int who = threadIdx.x;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(v, off);
                const int ow = __shfl_xor(who, off);
                const bool take = (ov < v) | ((ov == v) & (ow < who));
                v = take ? ov : v;
                who = take ? ow : who;
            }
            if (lane == 0) { red_v[wave] = v; red_t[wave] = who; }
            __syncthreads();
            double bv = red_v[0];
            int bt = red_t[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const bool take = (red_v[w] < bv) | ((red_v[w] == bv) & (red_t[w] < bt));
                bv = take ? red_v[w] : bv;
                bt = take ? red_t[w] : bt;
            }
            if ((int)threadIdx.x == bt) head += 1;
            if (threadIdx.x == 0) out[t] = bv;
            __syncthreads();
        }
        const double reach = (double)R * gr.h * (1.0 - 1e-9);
        if (whole || out[K - 1] <= reach * reach) break;         // block-uniform: `out` is shared
        if (R >= r_last) return;                                 // (block-uniform too) stays pending
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double res[KNN_KMAX];
#pragma unroll
        for (int j = 0; j < K; ++j) res[j] = sqrt(out[j]);
        mean_out[self] = numpy_pairwise_sum<K - 1>(res, 1) / (double)(K - 1);
        pending[self] = 0;
    }
}

template <int K>
hipError_t launch_box(const double *pts, const double *sorted, const Grid &gr, const int *start, int r_first, int r_last,
                      const int *queries, int n_queries, unsigned char *pending, double *mean_out, long long n, hipStream_t st)
{
    if (r_last <= r_first)        // the single-box pass: a wave per query
        hipLaunchKernelGGL((knn_box_kernel<K, 1>), dim3((unsigned)n_queries), dim3(64), 0, st, pts, sorted, gr, start, r_first,
                           r_last, queries, pending, mean_out, n);
    else
        hipLaunchKernelGGL((knn_box_kernel<K, 4>), dim3((unsigned)n_queries), dim3(256), 0, st, pts, sorted, gr, start, r_first,
                           r_last, queries, pending, mean_out, n);
    return hipGetLastError();
}

// number of cells holding at least one point
__global__ __launch_bounds__(256) void knn_occupied_kernel(const int *__restrict__ count, long long cells,
                                                           int *__restrict__ occupied)
{
    int local = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x)
        local += count[i] > 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(occupied, local);
}

// every stride-th point of a device-resident cloud, packed (the box estimate's sample)
__global__ __launch_bounds__(256) void knn_sample_kernel(const double *__restrict__ pts, long long stride, long long cnt,
                                                         double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt) return;
    out[3 * j] = pts[3 * j * stride];
    out[3 * j + 1] = pts[3 * j * stride + 1];
    out[3 * j + 2] = pts[3 * j * stride + 2];
}

__global__ __launch_bounds__(256) void knn_iota_kernel(int *__restrict__ v, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        v[i] = (int)i;
}

template <int K>
hipError_t launch_query(const double *sorted, long long n, const Grid &gr, const int *start, const int *origin,
                        unsigned char *pending, int max_shells, double *mean_out, hipStream_t st)
{
    hipLaunchKernelGGL((knn_query_kernel<K>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, sorted, n, gr,
                       start, origin, pending, max_shells, mean_out);
    return hipGetLastError();
}

}  // namespace

bool knn_supported(int k) { return k == 8 || k == 10 || k == 16 || k == 20 || k == 32; }

// points: [n][3] float64 on the host (or on the device: points_on_device); mean_out: host [n].  Needs n >= k.
hipError_t knn_mean_distance(const double *points, long long n, int k, double *mean_out, hipStream_t st,
                             bool points_on_device)
{
    // The grid spans the 1st .. 99th percentile of every axis (estimated on a strided sample); points
    // outside are binned into the border cells (per-axis clamping only under-estimates coordinate
    // differences, so the shell bound stays valid).  A handful of far outliers -- what this statistic
    // is computed to find -- would otherwise stretch the box until the body of the cloud falls into a
    // few crowded cells.
    double lo[3], hi[3];
    constexpr bool debug = AMVS_KNN_DEBUG;   // stage timings (development aid)
    auto now_ms = []() {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t_begin = debug ? now_ms() : 0.0;
    {
        const long long stride = std::max<long long>(1, n / 16384);
        const long long cnt = (n + stride - 1) / stride;
        // the strided sample on the host
        std::vector<double> sample((size_t)(3 * cnt));
        if (points_on_device) {
            // gathered on the device, then ONE contiguous copy (a strided hipMemcpy2D of 65 536 rows of 24 bytes took
            // 2 ms; dereferencing device memory from the host works through the PCIe BAR but takes ~4 us per read)
            double *d_sample = nullptr;
            KCHK(pool_malloc(&d_sample, sizeof(double) * 3 * (size_t)cnt));
            hipLaunchKernelGGL(knn_sample_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, points, stride, cnt, d_sample);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(sample.data(), d_sample, sizeof(double) * 3 * (size_t)cnt, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            pool_free(d_sample);
            if (e != hipSuccess) return e;
        } else {
            for (long long j = 0; j < cnt; ++j)
                for (int a = 0; a < 3; ++a) sample[(size_t)(3 * j + a)] = points[3 * j * stride + a];
        }
        std::vector<double> axis;
        axis.reserve((size_t)cnt);
        for (int a = 0; a < 3; ++a) {
            axis.clear();
            for (long long j = 0; j < cnt; ++j) axis.push_back(sample[(size_t)(3 * j + a)]);
            const size_t m = axis.size(), lo_k = (m - 1) / 100, hi_k = (m - 1) - lo_k;
            std::nth_element(axis.begin(), axis.begin() + lo_k, axis.end());
            lo[a] = axis[lo_k];
            std::nth_element(axis.begin(), axis.begin() + hi_k, axis.end());
            hi[a] = axis[hi_k];
        }
    }
    double ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) { ext[a] = hi[a] - lo[a]; emax = std::max(emax, ext[a]); }
    if (!(emax > 0.0) || !std::isfinite(emax)) emax = 1.0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], emax * 1e-3);
    // about 8 points per cell if the points filled the box, refined below from the occupancy actually
    // found (reconstructed clouds are surfaces plus diffuse noise, not volumes)
    double h = std::cbrt(vol * 8.0 / (double)n);
    const double h_min = emax / (KNN_GMAX - 2);      // keeps every in-box cell index below KNN_GMAX
    h = std::max(h, h_min);

    const double t_sampled = debug ? now_ms() : 0.0;
    double *d_pts = nullptr, *d_sorted = nullptr, *d_mean = nullptr;
    int *d_cell = nullptr, *d_count = nullptr, *d_start = nullptr, *d_origin = nullptr;
    unsigned char *d_pending = nullptr;
    int *d_occ = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    long long table_cap = 0;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_pts, (void *)d_sorted, (void *)d_mean, (void *)d_cell, (void *)d_count,
                        (void *)d_start, (void *)d_origin, (void *)d_pending, (void *)d_occ, d_tmp})
            if (p) pool_free(p);
    };
#define KCHK_C(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) { cleanup(); return e_; }                \
    } while (0)
    KCHK_C(pool_malloc(&d_pts, sizeof(double) * 3 * n));
    KCHK_C(pool_malloc(&d_sorted, sizeof(double) * 3 * n));
    KCHK_C(pool_malloc(&d_mean, sizeof(double) * n));
    KCHK_C(pool_malloc(&d_cell, sizeof(int) * n));
    KCHK_C(pool_malloc(&d_origin, sizeof(int) * n));
    KCHK_C(pool_malloc(&d_pending, (size_t)n));
    KCHK_C(pool_malloc(&d_occ, sizeof(int)));
    KCHK_C(hipMemcpyAsync(d_pts, points, sizeof(double) * 3 * n,
                          points_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    KCHK_C(hipMemsetAsync(d_pending, 1, (size_t)n, st));

    const int bx = (int)std::min<long long>((n + 255) / 256, 4096);
    Grid gr{};
    for (int a = 0; a < 3; ++a) gr.lo[a] = lo[a];
    long long cells = 0;
    // bins the points for edge `edge` (count + exclusive scan); returns points per occupied cell
    auto bin = [&](double edge, double &per_cell) -> hipError_t {
        gr.h = edge; gr.inv_h = 1.0 / edge;
        for (int a = 0; a < 3; ++a) gr.g[a] = std::max(1, std::min(KNN_GMAX, (int)std::floor(ext[a] / edge) + 1));
        cells = (long long)gr.g[0] * gr.g[1] * gr.g[2];
        if (cells + 1 > table_cap) {
            if (d_count) pool_free(d_count);
            if (d_start) pool_free(d_start);
            d_count = d_start = nullptr;
            KCHK(pool_malloc(&d_count, sizeof(int) * (cells + 1)));
            KCHK(pool_malloc(&d_start, sizeof(int) * (cells + 1)));
            table_cap = cells + 1;
        }
        KCHK(hipMemsetAsync(d_count, 0, sizeof(int) * (cells + 1), st));
        hipLaunchKernelGGL(knn_count_kernel, dim3(bx), dim3(256), 0, st, d_pts, n, gr, d_cell, d_count);
        KCHK(hipGetLastError());
        size_t need = 0;
        KCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_count, d_start, (int)(cells + 1), st));
        if (need > tmp_bytes) {
            if (d_tmp) pool_free(d_tmp);
            d_tmp = nullptr;
            KCHK(pool_malloc(&d_tmp, need));
            tmp_bytes = need;
        }
        KCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_count, d_start, (int)(cells + 1), st));
        if (per_cell >= 0.0) {
            KCHK(hipMemsetAsync(d_occ, 0, sizeof(int), st));
            const int cb = (int)std::min<long long>((cells + 255) / 256, 2048);
            hipLaunchKernelGGL(knn_occupied_kernel, dim3(cb), dim3(256), 0, st, d_count, cells, d_occ);
            KCHK(hipGetLastError());
            int occupied = 0;
            KCHK(hipMemcpyAsync(&occupied, d_occ, sizeof(int), hipMemcpyDeviceToHost, st));
            KCHK(hipStreamSynchronize(st));
            per_cell = (double)n / (double)std::max(occupied, 1);
        }
        return hipSuccess;
    };
    auto query = [&](int max_shells) -> hipError_t {
        KCHK(hipMemsetAsync(d_count, 0, sizeof(int) * (cells + 1), st));     // reused as the placement cursor
        hipLaunchKernelGGL(knn_place_kernel, dim3(bx), dim3(256), 0, st, d_pts, n, d_cell, d_start, d_count,
                           d_sorted, d_origin, cells);
        KCHK(hipGetLastError());
        switch (k) {
        case 8: return launch_query<8>(d_sorted, n, gr, d_start, d_origin, d_pending, max_shells, d_mean, st);
        case 10: return launch_query<10>(d_sorted, n, gr, d_start, d_origin, d_pending, max_shells, d_mean, st);
        case 16: return launch_query<16>(d_sorted, n, gr, d_start, d_origin, d_pending, max_shells, d_mean, st);
        case 20: return launch_query<20>(d_sorted, n, gr, d_start, d_origin, d_pending, max_shells, d_mean, st);
        case 32: return launch_query<32>(d_sorted, n, gr, d_start, d_origin, d_pending, max_shells, d_mean, st);
        default: return hipErrorInvalidValue;
        }
    };

    // level 0: edge fitted to the occupancy of the occupied cells (3 .. 24 points)
    double fitted_per_cell = 0.0;
    int walk_again = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        double per_cell = 0.0;
        KCHK_C(bin(h, per_cell));
        fitted_per_cell = per_cell;
        const bool at_limit = gr.g[0] == KNN_GMAX || gr.g[1] == KNN_GMAX || gr.g[2] == KNN_GMAX;
        if (per_cell > 24.0 && !at_limit && h > h_min) { h = std::max(h * 0.5, h_min); continue; }
        if (per_cell < 3.0 && cells > 1) { h *= 2.0; continue; }
        break;
    }
    // coarser levels (edge x2 each) pick up the queries whose neighbourhood is sparser than two
    // shells of the level before; the last level may scan
    const double t_binned = debug ? ((void)hipStreamSynchronize(st), now_ms()) : 0.0;
    int *d_iota = nullptr, *d_queries = nullptr, *d_nsel = nullptr;
    auto cleanup2 = [&]() {
        for (void *p : {(void *)d_iota, (void *)d_queries, (void *)d_nsel})
            if (p) pool_free(p);
        d_iota = d_queries = d_nsel = nullptr;
    };
#define KCHK_D(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) { cleanup2(); cleanup(); return e_; }    \
    } while (0)
    KCHK_D(pool_malloc(&d_iota, sizeof(int) * n));
    KCHK_D(pool_malloc(&d_queries, sizeof(int) * n));
    KCHK_D(pool_malloc(&d_nsel, sizeof(int)));
    hipLaunchKernelGGL(knn_iota_kernel, dim3(bx), dim3(256), 0, st, d_iota, n);
    KCHK_D(hipGetLastError());
    // the still-pending queries, compacted (and counted) after every level
    auto pending_list = [&](int &count) -> hipError_t {
        size_t need = 0;
        KCHK(hipcub::DeviceSelect::Flagged(nullptr, need, d_iota, d_pending, d_queries, d_nsel, (int)n, st));
        if (need > tmp_bytes) {
            if (d_tmp) pool_free(d_tmp);
            d_tmp = nullptr;
            KCHK(pool_malloc(&d_tmp, need));
            tmp_bytes = need;
        }
        KCHK(hipcub::DeviceSelect::Flagged(d_tmp, tmp_bytes, d_iota, d_pending, d_queries, d_nsel, (int)n, st));
        KCHK(hipMemcpyAsync(&count, d_nsel, sizeof(int), hipMemcpyDeviceToHost, st));
        return hipStreamSynchronize(st);
    };
    // the cell walk on the fine grid, then one cooperative box scan per query it left pending
    int left = (int)n;
    {
        hipEvent_t e0, e1, e2;
        if (debug) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2); (void)hipEventRecord(e0, st); }
        auto box = [&](int r_first, int r_last, int count) -> hipError_t {
            switch (k) {
            case 8: return launch_box<8>(d_pts, d_sorted, gr, d_start, r_first, r_last, d_queries, count, d_pending, d_mean, n, st);
            case 10: return launch_box<10>(d_pts, d_sorted, gr, d_start, r_first, r_last, d_queries, count, d_pending, d_mean, n, st);
            case 16: return launch_box<16>(d_pts, d_sorted, gr, d_start, r_first, r_last, d_queries, count, d_pending, d_mean, n, st);
            case 20: return launch_box<20>(d_pts, d_sorted, gr, d_start, r_first, r_last, d_queries, count, d_pending, d_mean, n, st);
            case 32: return launch_box<32>(d_pts, d_sorted, gr, d_start, r_first, r_last, d_queries, count, d_pending, d_mean, n, st);
            default: return hipErrorInvalidValue;
            }
        };
        // a thread per query walks shells 0..1; a wave per query left scans the box R = 2 (the cells of shells
        // 0..2); a block per query still left scans the boxes R = 4, 8, ... until its rule holds
        KCHK_D(query(1));
        KCHK_D(pending_list(left));
        walk_again = left;
        if (left > 0) {
            KCHK_D(box(KNN_SHELLS, KNN_SHELLS, left));
            KCHK_D(pending_list(left));
        }
        if (debug) (void)hipEventRecord(e1, st);
        if (left > 0) KCHK_D(box(2 * KNN_SHELLS, 1 << 30, left));     // (R = 4 as a wave's pass of its own: no gain, measured)
        if (debug) {
            (void)hipEventRecord(e2, st);
            (void)hipStreamSynchronize(st);
            float ms0 = 0.f, ms1 = 0.f;
            (void)hipEventElapsedTime(&ms0, e0, e1);
            (void)hipEventElapsedTime(&ms1, e1, e2);
            std::fprintf(stderr, "knn: h %.4g grid %dx%dx%d (%.1f points per occupied cell)  cell walk + box 2: %.2f ms (%d queries to the box), "
                         "%d of %lld pending -> boxes 4, 8, ...: %.2f ms\n", gr.h, gr.g[0], gr.g[1], gr.g[2], fitted_per_cell, ms0, walk_again,
                         left, n, ms1);
        }
    }
    const double t_searched = debug ? ((void)hipStreamSynchronize(st), now_ms()) : 0.0;
    cleanup2();
#undef KCHK_D
    KCHK_C(hipMemcpyAsync(mean_out, d_mean, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    KCHK_C(hipStreamSynchronize(st));
    const double t_copied = debug ? now_ms() : 0.0;
    cleanup();
    if (debug)
        std::fprintf(stderr, "knn: box sample %.2f ms, allocate + bin %.2f, search %.2f, result to the host %.2f, release %.2f\n",
                     t_sampled - t_begin, t_binned - t_sampled, t_searched - t_binned, t_copied - t_searched, now_ms() - t_copied);
    return hipSuccess;
#undef KCHK_C
}

}  // namespace amvs

AMVS_CHECK_TU(knn)
