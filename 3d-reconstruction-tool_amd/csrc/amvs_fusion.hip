// amvs_fusion.hip -- device-side fusion and filtering of the per-view maps into a point cloud
// (reference: PatchMatchMVS._fuse_depth_maps mvs_patchmatch.py:536-570 and _filter_points
// :572-588, both float64 NumPy on the host there).  Same results bit for bit: float64 throughout,
// 3-term products as the left-to-right FMA chain NumPy's matmul uses, order-preserving stream
// compaction (np.where order), np.median / np.percentile(95, linear) on device-sorted columns,
// voxel de-duplication as a stable radix sort of the int64 keys keeping the first point of every
// run (np.unique(return_index=True)).  hipCUB supplies the scans / sorts / selects.
#define AMVS_TU_ID 7
#include "amvs_check.h"
#include "amvs_kernels.h"
#include "amvs_pool.h"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <vector>

namespace amvs {

namespace {

#define FCHK(call)                                 \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) return e_;           \
    } while (0)

// per-pixel flag: confidence >= min_views (mvs_patchmatch.py:545)
__global__ __launch_bounds__(256) void fuse_flag_kernel(const float *__restrict__ conf, long long n,
                                                        float min_views, unsigned char *__restrict__ flag)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        flag[i] = conf[i] >= min_views ? 1 : 0;
}

// stereo: confidence >= min_confidence and depth > 0 (dense_stereo.py:414)
__global__ __launch_bounds__(256) void stereo_flag_kernel(const float *__restrict__ conf, const float *__restrict__ depth,
                                                          long long n, float min_conf, unsigned char *__restrict__ flag)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        flag[i] = ((conf[i] >= min_conf) & (depth[i] > 0.0f)) ? 1 : 0;
}

// first[j] = number of selected (ascending) indices below j*HW, j = 0 .. n_maps  (lower bounds)
__global__ void map_bounds_kernel(const long long *__restrict__ sel, long long m, long long HW, int n_maps,
                                  long long *__restrict__ first)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_maps) return;
    const long long key = (long long)j * HW;
    long long lo = 0, hi = m;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sel[mid] < key) lo = mid + 1; else hi = mid;
    }
    first[j] = lo;
}

// back-project the selected pixels: rays = [x,y,1] @ K_inv.T; X = rays*d; Xw = (X - t) @ R
// (mvs_patchmatch.py:556-562), all float64
__global__ __launch_bounds__(256) void fuse_project_kernel(const long long *__restrict__ sel, long long m,
                                                           const float *__restrict__ depth,
                                                           const unsigned char *__restrict__ bgr, int H, int W,
                                                           const double *__restrict__ Kinv,
                                                           const double *__restrict__ poses,   // [n_maps][12]: R row-major, t
                                                           int n_maps, double *__restrict__ pts, unsigned char *__restrict__ rgb)
{
    const long long HW = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        const long long g = AMVS_IDX(sel[i], (long long)n_maps * HW);        // (pixel of the stacked maps)
        const int map = (int)(g / HW);
        const long long p = g - map * HW;
        const double y = (double)(p / W), x = (double)(p % W);
        const double d = (double)depth[g];
        const double *R = poses + 12 * map, *t = R + 9;
        double c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ray = fma(1.0, Kinv[3 * j + 2], fma(y, Kinv[3 * j + 1], x * Kinv[3 * j]));
            c[j] = ray * d - t[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) pts[3 * i + j] = fma(c[2], R[6 + j], fma(c[1], R[3 + j], c[0] * R[j]));
        rgb[3 * i] = bgr[3 * g + 2]; rgb[3 * i + 1] = bgr[3 * g + 1]; rgb[3 * i + 2] = bgr[3 * g];   // BGR -> RGB (:565)
    }
}

__global__ __launch_bounds__(256) void column_kernel(const double *__restrict__ pts, long long m, int col,
                                                     double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = pts[3 * i + col];
}

// np.linalg.norm(points - centroid, axis=1): sqrt((dx*dx + dy*dy) + dz*dz)
__global__ __launch_bounds__(256) void dist_kernel(const double *__restrict__ pts, long long m, double cx, double cy,
                                                   double cz, double *__restrict__ dist)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        const double dx = pts[3 * i] - cx, dy = pts[3 * i + 1] - cy, dz = pts[3 * i + 2] - cz;
        dist[i] = sqrt(dx * dx + dy * dy + dz * dz);
    }
}

__global__ __launch_bounds__(256) void below_kernel(const double *__restrict__ dist, long long m, double thr,
                                                    unsigned char *__restrict__ flag)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x)
        flag[i] = dist[i] < thr ? 1 : 0;
}

// voxel key of mvs_patchmatch.py:583-586 for the points selected by `sel`
__global__ __launch_bounds__(256) void voxel_key_kernel(const double *__restrict__ pts, const long long *__restrict__ sel,
                                                        long long m, long long m_src, double voxel, long long *__restrict__ keys)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        const long long s = AMVS_IDX(sel[i], m_src);                          // (point of the source cloud)
        const long long ix = (long long)floor(pts[3 * s] / voxel);
        const long long iy = (long long)floor(pts[3 * s + 1] / voxel);
        const long long iz = (long long)floor(pts[3 * s + 2] / voxel);
        keys[i] = ix * 1000000000ll + iy * 1000000ll + iz;
    }
}

__global__ __launch_bounds__(256) void run_head_kernel(const long long *__restrict__ keys, long long m,
                                                       unsigned char *__restrict__ flag)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x)
        flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void iota_kernel(long long *__restrict__ out, long long m)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = i;
}

// out[i] = src[map[idx[i]]] for points and colours
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ pts, const unsigned char *__restrict__ rgb,
                                                     const long long *__restrict__ sel, const long long *__restrict__ pick,
                                                     long long m, long long m_sel, long long m_src, double *__restrict__ pts_out,
                                                     unsigned char *__restrict__ rgb_out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        const long long s = AMVS_IDX(sel[AMVS_IDX(pick[i], m_sel)], m_src);
#pragma unroll
        for (int j = 0; j < 3; ++j) { pts_out[3 * i + j] = pts[3 * s + j]; rgb_out[3 * i + j] = rgb[3 * s + j]; }
    }
}

// out[i] = src[idx[i]] for points and colours: the rows `idx` of a cloud, in that order
__global__ __launch_bounds__(256) void take_kernel(const double *__restrict__ pts, const unsigned char *__restrict__ rgb,
                                                   const long long *__restrict__ idx, long long m, long long n_src,
                                                   double *__restrict__ pts_out, unsigned char *__restrict__ rgb_out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        const long long s = AMVS_IDX(idx[i], n_src);
#pragma unroll
        for (int j = 0; j < 3; ++j) { pts_out[3 * i + j] = pts[3 * s + j]; rgb_out[3 * i + j] = rgb[3 * s + j]; }
    }
}

inline dim3 grid_for(long long n) { long long b = (n + 255) / 256; return dim3((unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b))); }

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) pool_free(p);
        p = nullptr; cap = 0;
        hipError_t e = pool_malloc(&p, n);                  // (amvs_pool.hip: blocks cached between calls)
        if (e == hipSuccess) cap = n;
        return e;
    }
    ~Scratch() { if (p) pool_free(p); }
};

// order-preserving selection of the indices [0,n) whose flag is set
hipError_t select_indices(const unsigned char *flag, long long n, long long *out, long long *d_count,
                          long long *h_count, Scratch &tmp, hipStream_t st)
{
    hipcub::CountingInputIterator<long long> iota(0);
    size_t bytes = 0;
    FCHK(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota, flag, out, d_count, (int)n, st));
    FCHK(tmp.need(bytes));
    FCHK(hipcub::DeviceSelect::Flagged(tmp.p, bytes, iota, flag, out, d_count, (int)n, st));
    FCHK(hipMemcpyAsync(h_count, d_count, sizeof(long long), hipMemcpyDeviceToHost, st));
    return hipStreamSynchronize(st);
}

hipError_t sort_keys(double *in, double *out, long long n, Scratch &tmp, hipStream_t st)
{
    size_t bytes = 0;
    FCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, in, out, (int)n, 0, 64, st));
    FCHK(tmp.need(bytes));
    return hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, in, out, (int)n, 0, 64, st);
}

// First point of every voxel in key order (np.unique(keys, return_index=True)) among the points
// sel[0 .. m2) of (pts, rgb): stable radix sort of the int64 keys, heads of the runs, gather.
hipError_t voxel_first_of_key(const double *pts, const unsigned char *rgb, long long m_src, const long long *sel, long long m2,
                              double voxel, Scratch &tmp, Scratch &flag, Scratch &cnt, double **pts2_out,
                              unsigned char **rgb2_out, long long *m3_out, hipStream_t st)
{
    *pts2_out = nullptr; *rgb2_out = nullptr; *m3_out = 0;
    Scratch keysA, keysB, idxA, idxB, pick;
    FCHK(keysA.need(8 * m2)); FCHK(keysB.need(8 * m2)); FCHK(idxA.need(8 * m2)); FCHK(idxB.need(8 * m2));
    FCHK(pick.need(8 * m2));
    FCHK(flag.need(m2));
    hipLaunchKernelGGL(voxel_key_kernel, grid_for(m2), dim3(256), 0, st, pts, sel, m2, m_src, voxel, (long long *)keysA.p);
    hipLaunchKernelGGL(iota_kernel, grid_for(m2), dim3(256), 0, st, (long long *)idxA.p, m2);
    {
        size_t bytes = 0;
        FCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (long long *)keysA.p, (long long *)keysB.p,
                                                (long long *)idxA.p, (long long *)idxB.p, (int)m2, 0, 64, st));
        FCHK(tmp.need(bytes));
        FCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, (long long *)keysA.p, (long long *)keysB.p,
                                                (long long *)idxA.p, (long long *)idxB.p, (int)m2, 0, 64, st));
    }
    hipLaunchKernelGGL(run_head_kernel, grid_for(m2), dim3(256), 0, st, (const long long *)keysB.p, m2, (unsigned char *)flag.p);
    long long m3 = 0;
    {   // positions (in sorted order) of the run heads -> original indices idxB[pos]
        size_t bytes = 0;
        FCHK(hipcub::DeviceSelect::Flagged(nullptr, bytes, (long long *)idxB.p, (unsigned char *)flag.p,
                                           (long long *)pick.p, (long long *)cnt.p, (int)m2, st));
        FCHK(tmp.need(bytes));
        FCHK(hipcub::DeviceSelect::Flagged(tmp.p, bytes, (long long *)idxB.p, (unsigned char *)flag.p,
                                           (long long *)pick.p, (long long *)cnt.p, (int)m2, st));
        FCHK(hipMemcpyAsync(&m3, cnt.p, 8, hipMemcpyDeviceToHost, st));
        FCHK(hipStreamSynchronize(st));
    }
    double *pts2 = nullptr;
    unsigned char *rgb2 = nullptr;
    FCHK(hipMalloc(&pts2, sizeof(double) * 3 * (m3 > 0 ? m3 : 1)));
    hipError_t e = hipMalloc(&rgb2, 3 * (m3 > 0 ? m3 : 1));
    if (e != hipSuccess) { (void)hipFree(pts2); return e; }
    hipLaunchKernelGGL(gather_kernel, grid_for(m3), dim3(256), 0, st, pts, rgb, sel, (const long long *)pick.p, m3, m2, m_src, pts2, rgb2);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(pts2); (void)hipFree(rgb2); return e; }
    *pts2_out = pts2; *rgb2_out = rgb2; *m3_out = m3;
    return hipSuccess;
}

// np.median of a sorted column
double median_sorted(const std::vector<double> &mid, long long n) { return n % 2 ? mid[0] : (mid[0] + mid[1]) / 2.0; }

}  // namespace

// Fusion (+ optional filter).  depth/conf: [n_maps][H*W] float32 on the device; bgr: [n_maps][H*W][3]
// uint8 on the device; Kinv: 9 doubles, poses: n_maps x 12 doubles (host).  Results stay on the
// device in *pts_out / *rgb_out (hipMalloc'ed here, owned by the caller); counts[0] = raw points,
// counts[1] = points after the filter.
hipError_t fuse_filter(const float *depth, const float *conf, const unsigned char *bgr, int n_maps, int H, int W,
                       const double *Kinv_h, const double *poses_h, float min_views, bool do_filter,
                       double **pts_out, unsigned char **rgb_out, long long counts[2], hipStream_t st)
{
    *pts_out = nullptr; *rgb_out = nullptr; counts[0] = counts[1] = 0;
    const long long n = (long long)n_maps * H * W;
    if (n <= 0 || n > 0x7FFFFFFFll) return hipErrorInvalidValue;
    Scratch tmp, flag, sel, consts, cnt;
    FCHK(flag.need(n));
    FCHK(sel.need(sizeof(long long) * n));
    FCHK(consts.need(sizeof(double) * (9 + 12 * n_maps)));
    FCHK(cnt.need(sizeof(long long)));
    double *d_Kinv = (double *)consts.p, *d_poses = d_Kinv + 9;
    FCHK(hipMemcpyAsync(d_Kinv, Kinv_h, sizeof(double) * 9, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_poses, poses_h, sizeof(double) * 12 * n_maps, hipMemcpyHostToDevice, st));

    // ---- fusion: np.where(confidence >= min_views), view by view, row-major ----
    hipLaunchKernelGGL(fuse_flag_kernel, grid_for(n), dim3(256), 0, st, conf, n, min_views, (unsigned char *)flag.p);
    long long m = 0;
    FCHK(select_indices((unsigned char *)flag.p, n, (long long *)sel.p, (long long *)cnt.p, &m, tmp, st));
    counts[0] = counts[1] = m;
    if (m == 0) return hipSuccess;
    double *pts = nullptr;
    unsigned char *rgb = nullptr;
    FCHK(hipMalloc(&pts, sizeof(double) * 3 * m));
    hipError_t e = hipMalloc(&rgb, 3 * m);
    if (e != hipSuccess) { (void)hipFree(pts); return e; }
    hipLaunchKernelGGL(fuse_project_kernel, grid_for(m), dim3(256), 0, st, (const long long *)sel.p, m, depth, bgr, H,
                       W, d_Kinv, d_poses, n_maps, pts, rgb);
    auto bail = [&](hipError_t err) { (void)hipFree(pts); (void)hipFree(rgb); return err; };
    if (!do_filter) {
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return bail(e);
        *pts_out = pts; *rgb_out = rgb;
        return hipSuccess;
    }

    // ---- filter: 95th-percentile radius around the per-axis median ----
    Scratch colA, colB;
    if ((e = colA.need(sizeof(double) * m)) != hipSuccess) return bail(e);
    if ((e = colB.need(sizeof(double) * m)) != hipSuccess) return bail(e);
    double centroid[3];
    const long long lo = (m - 1) / 2;                         // middle element(s) of a sorted column
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(column_kernel, grid_for(m), dim3(256), 0, st, pts, m, c, (double *)colA.p);
        if ((e = sort_keys((double *)colA.p, (double *)colB.p, m, tmp, st)) != hipSuccess) return bail(e);
        std::vector<double> mid(2, 0.0);
        if ((e = hipMemcpyAsync(mid.data(), (double *)colB.p + lo, sizeof(double) * (m % 2 ? 1 : 2),
                                hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return bail(e);
        centroid[c] = median_sorted(mid, m);
    }
    hipLaunchKernelGGL(dist_kernel, grid_for(m), dim3(256), 0, st, pts, m, centroid[0], centroid[1], centroid[2],
                       (double *)colA.p);
    if ((e = sort_keys((double *)colA.p, (double *)colB.p, m, tmp, st)) != hipSuccess) return bail(e);
    // np.percentile(d, 95), method 'linear': virtual index 0.95*(m-1), numpy's _lerp
    double thr;
    {
        const double vi = (95.0 / 100.0) * (double)(m - 1);
        long long prev = (long long)std::floor(vi);
        if (prev > m - 1) prev = m - 1;
        const long long next = prev + 1 < m ? prev + 1 : m - 1;
        const double t = vi - (double)prev;
        double ab[2];
        if ((e = hipMemcpyAsync(&ab[0], (double *)colB.p + prev, sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e);
        if ((e = hipMemcpyAsync(&ab[1], (double *)colB.p + next, sizeof(double), hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return bail(e);
        const double diff = ab[1] - ab[0];
        thr = t >= 0.5 ? ab[1] - diff * (1.0 - t) : ab[0] + diff * t;
    }
    hipLaunchKernelGGL(below_kernel, grid_for(m), dim3(256), 0, st, (const double *)colA.p, m, thr, (unsigned char *)flag.p);
    long long m2 = 0;
    if ((e = select_indices((unsigned char *)flag.p, m, (long long *)sel.p, (long long *)cnt.p, &m2, tmp, st)) != hipSuccess) return bail(e);
    if (m2 == 0) { (void)hipFree(pts); (void)hipFree(rgb); counts[1] = 0; return hipSuccess; }

    // ---- voxel de-duplication: first point of every key, in key order ----
    double *pts2 = nullptr;
    unsigned char *rgb2 = nullptr;
    long long m3 = 0;
    e = voxel_first_of_key(pts, rgb, m, (const long long *)sel.p, m2, 0.01, tmp, flag, cnt, &pts2, &rgb2, &m3, st);
    (void)hipFree(pts); (void)hipFree(rgb);
    if (e != hipSuccess) return e;
    *pts_out = pts2; *rgb_out = rgb2;
    counts[1] = m3;
    return hipSuccess;
}

// DenseStereoReconstructor._backproject (dense_stereo.py:407-437) for n_maps reference views at once:
// pixels with confidence >= min_confidence and depth > 0, view by view in row-major order, through
// the same float64 chain as fuse_project_kernel (float32 pixel coordinates convert exactly).
// per_map_h[j] (host, optional) = points of map j -- the counts of the reference's progress lines.
hipError_t stereo_backproject(const float *depth, const float *conf, const unsigned char *bgr, int n_maps, int H, int W,
                              const double *Kinv_h, const double *poses_h, float min_confidence, double **pts_out,
                              unsigned char **rgb_out, long long *total, long long *per_map_h, hipStream_t st)
{
    *pts_out = nullptr; *rgb_out = nullptr; *total = 0;
    const long long HW = (long long)H * W, n = (long long)n_maps * HW;
    if (n <= 0 || n > 0x7FFFFFFFll) return hipErrorInvalidValue;
    Scratch tmp, flag, sel, consts, cnt, bounds;
    FCHK(flag.need(n));
    FCHK(sel.need(sizeof(long long) * n));
    FCHK(consts.need(sizeof(double) * (9 + 12 * n_maps)));
    FCHK(cnt.need(sizeof(long long)));
    FCHK(bounds.need(sizeof(long long) * (n_maps + 1)));
    double *d_Kinv = (double *)consts.p, *d_poses = d_Kinv + 9;
    FCHK(hipMemcpyAsync(d_Kinv, Kinv_h, sizeof(double) * 9, hipMemcpyHostToDevice, st));
    FCHK(hipMemcpyAsync(d_poses, poses_h, sizeof(double) * 12 * n_maps, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(stereo_flag_kernel, grid_for(n), dim3(256), 0, st, conf, depth, n, min_confidence,
                       (unsigned char *)flag.p);
    long long m = 0;
    FCHK(select_indices((unsigned char *)flag.p, n, (long long *)sel.p, (long long *)cnt.p, &m, tmp, st));
    *total = m;
    if (per_map_h) {
        std::vector<long long> first(n_maps + 1, 0);
        hipLaunchKernelGGL(map_bounds_kernel, dim3((n_maps + 1 + 63) / 64), dim3(64), 0, st, (const long long *)sel.p, m, HW,
                           n_maps, (long long *)bounds.p);
        FCHK(hipMemcpyAsync(first.data(), bounds.p, sizeof(long long) * (n_maps + 1), hipMemcpyDeviceToHost, st));
        FCHK(hipStreamSynchronize(st));
        for (int j = 0; j < n_maps; ++j) per_map_h[j] = first[j + 1] - first[j];
    }
    if (m == 0) return hipSuccess;
    double *pts = nullptr;
    unsigned char *rgb = nullptr;
    FCHK(hipMalloc(&pts, sizeof(double) * 3 * m));
    hipError_t e = hipMalloc(&rgb, 3 * m);
    if (e != hipSuccess) { (void)hipFree(pts); return e; }
    hipLaunchKernelGGL(fuse_project_kernel, grid_for(m), dim3(256), 0, st, (const long long *)sel.p, m, depth, bgr, H,
                       W, d_Kinv, d_poses, n_maps, pts, rgb);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(pts); (void)hipFree(rgb); return e; }
    *pts_out = pts; *rgb_out = rgb;
    return hipSuccess;
}

// points[chosen], colors[chosen] of a device cloud (dense_stereo.py:449-455: the random sub-sample of clouds above
// 500 000 points; the caller draws `chosen` with numpy as the reference does): rows idx_h[0 .. m) in that order
hipError_t cloud_take(const double *pts, const unsigned char *rgb, long long n, const long long *idx_h, long long m,
                      double **pts_out, unsigned char **rgb_out, hipStream_t st)
{
    *pts_out = nullptr; *rgb_out = nullptr;
    if (m <= 0) return hipSuccess;
    for (long long i = 0; i < m; ++i)
        if (idx_h[i] < 0 || idx_h[i] >= n) return hipErrorInvalidValue;
    Scratch idx;
    FCHK(idx.need(sizeof(long long) * m));
    FCHK(hipMemcpyAsync(idx.p, idx_h, sizeof(long long) * m, hipMemcpyHostToDevice, st));
    double *pts2 = nullptr;
    unsigned char *rgb2 = nullptr;
    FCHK(hipMalloc(&pts2, sizeof(double) * 3 * m));
    hipError_t e = hipMalloc(&rgb2, 3 * m);
    if (e != hipSuccess) { (void)hipFree(pts2); return e; }
    hipLaunchKernelGGL(take_kernel, grid_for(m), dim3(256), 0, st, pts, rgb, (const long long *)idx.p, m, n, pts2, rgb2);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(pts2); (void)hipFree(rgb2); return e; }
    *pts_out = pts2; *rgb_out = rgb2;
    return hipSuccess;
}

// DenseStereoReconstructor._voxel_down_sample (dense_stereo.py:475-492) of a device cloud, after an
// optional keep mask (host, m bytes; the outlier filter's selection): first point of every voxel in
// key order.  The reference casts the voxel indices through int32 before forming the int64 key; for
// clouds inside +-2^31 voxels that is the identity.
hipError_t voxel_downsample(const double *pts, const unsigned char *rgb, long long m, const unsigned char *keep_h,
                            double voxel, double **pts_out, unsigned char **rgb_out, long long *m_out, hipStream_t st)
{
    *pts_out = nullptr; *rgb_out = nullptr; *m_out = 0;
    if (m <= 0) return hipSuccess;
    if (m > 0x7FFFFFFFll) return hipErrorInvalidValue;
    Scratch tmp, flag, sel, cnt;
    FCHK(flag.need(m));
    FCHK(sel.need(sizeof(long long) * m));
    FCHK(cnt.need(sizeof(long long)));
    long long m2 = m;
    if (keep_h) {
        FCHK(hipMemcpyAsync(flag.p, keep_h, m, hipMemcpyHostToDevice, st));
        FCHK(select_indices((unsigned char *)flag.p, m, (long long *)sel.p, (long long *)cnt.p, &m2, tmp, st));
    } else {
        hipLaunchKernelGGL(iota_kernel, grid_for(m), dim3(256), 0, st, (long long *)sel.p, m);
    }
    if (m2 == 0) return hipSuccess;
    return voxel_first_of_key(pts, rgb, m, (const long long *)sel.p, m2, voxel, tmp, flag, cnt, pts_out, rgb_out, m_out, st);
}

}  // namespace amvs

AMVS_CHECK_TU(fusion)
