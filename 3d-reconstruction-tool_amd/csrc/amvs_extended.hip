// amvs_extended.hip -- the EXTENDED PatchMatch mode: what the reference's
// module docstring names (src/core/mvs_patchmatch.py:1-13: plane hypotheses with normals, spatial
// propagation, VIEW propagation, random refinement) but its code does not implement -- the reference
// ignores the normal in the cost (:323-390), pulls whole shifted maps Jacobi-style (:415-457) and has no
// view propagation at all.  There is no reference counterpart, hence no parity: this mode is flagged
// (off by default) and judged against synthetic ground-truth depth (tests/test_extended_mode.py).
//
//   cost        slanted-plane homography: hypothesis (d, n) at pixel p defines the plane n.X = n.(d K^-1 p);
//               every window pixel q is lifted onto that plane (t_q = delta / n.K^-1 q), projected into
//               each source with the precomposed map of the fast mode ([uz,vz,z] = t_q (M q) + b) and
//               sampled bilinearly; NCC per source over the window samples; the cost is the mean of the
//               better half of the valid sources (occlusion-robust), +inf with fewer than two.
//   schedule    red-black: the pixels of one checkerboard colour test the planes of their four
//               neighbours of the other colour (extended to their own ray), in place -- Gauss-Seidel
//               instead of the reference's Jacobi pulls.
//   view prop.  before each iteration every pixel receives one candidate from a source view's current
//               map (a SNAPSHOT: no map is written during that pass, so results are deterministic): the
//               source's local plane at the pixel's projection, carried into the reference frame and
//               intersected with the pixel's ray.  On several GPUs this is what the all-gather of the
//               depth / normal maps between sweeps feeds.
//   refinement  per iteration: shrinking relative depth / normal perturbations and, early on, a fresh
//               random hypothesis.
//   consistency number of sources whose own depth map agrees after forward-backward reprojection
//               (pixel error < 1, relative depth error < 1 %); fusion keeps pixels with enough of them.
#define AMVS_TU_ID 6
#include "amvs_kernel_common.h"

namespace amvs {

// job-table fields are read through the constant address space (scalar loads)
typedef const __attribute__((address_space(4))) float *CF;

AMVS_DEV float xbilinear(const float *img, int H, int W, float u, float v, bool &inside)
{
    const float x0f = __builtin_floorf(u), y0f = __builtin_floorf(v);
    inside = (x0f >= 0.0f) & (y0f >= 0.0f) & (x0f <= (float)(W - 2)) & (y0f <= (float)(H - 2));
    if (!inside) return 0.0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float fx = u - x0f, fy = v - y0f;
    const float *p = img + AMVS_IDX((long long)y0 * W + x0, (long long)H * W - W - 1);   // (reads p[0], p[1], p[W], p[W + 1])
    const float top = __builtin_fmaf(fx, p[1] - p[0], p[0]);
    const float bot = __builtin_fmaf(fx, p[W + 1] - p[W], p[W]);
    return __builtin_fmaf(fy, bot - top, top);
}

// ascending list of per-source costs in registers (constant indices only: no scratch memory)
AMVS_DEV void xsorted_insert(float (&costs)[AMVS_KMAX_SRC], float c)
{
#pragma unroll
    for (int j = 0; j < AMVS_KMAX_SRC; ++j) {
        const float lo = __builtin_fminf(costs[j], c), hi = __builtin_fmaxf(costs[j], c);
        costs[j] = lo; c = hi;
    }
}

// mean of the better half of the valid sources (at least two), +inf with fewer than two
AMVS_DEV float xbetter_half(const float (&costs)[AMVS_KMAX_SRC], int n_valid)
{
    if (n_valid < 2) return __builtin_inff();
    const int keep = (n_valid + 1) / 2 > 2 ? (n_valid + 1) / 2 : 2;
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < AMVS_KMAX_SRC; ++j) tot += j < keep ? costs[j] : 0.0f;
    return tot / (float)keep;
}

// cost of hypothesis (d, n) at pixel (x, y) of the job's reference view
AMVS_DEV float xcost(const XArgs &a, JobCP job, const float *ref, int x, int y, float d, float nx, float ny, float nz)
{
    const int H = a.H, W = a.W, half = a.patch / 2;
    const float k0 = job->Kinv[0], k1 = job->Kinv[1], k2 = job->Kinv[2], k3 = job->Kinv[3], k4 = job->Kinv[4],
                k5 = job->Kinv[5];
    // ray r_q = K^-1 [qx, qy, 1] (third row of K^-1 assumed [0 0 1]); n.r_q is affine in q
    const float rpx = __builtin_fmaf(k1, (float)y, __builtin_fmaf(k0, (float)x, k2));
    const float rpy = __builtin_fmaf(k4, (float)y, __builtin_fmaf(k3, (float)x, k5));
    const float ndr_p = __builtin_fmaf(nx, rpx, __builtin_fmaf(ny, rpy, nz));
    const float delta = d * ndr_p;                            // plane: n.X = delta
    if (!(ndr_p < -1e-6f)) return __builtin_inff();           // plane must face the camera
    float costs[AMVS_KMAX_SRC];
#pragma unroll
    for (int j = 0; j < AMVS_KMAX_SRC; ++j) costs[j] = __builtin_inff();
    int n_valid = 0;
    for (int s = 0; s < a.n_src; ++s) {
        const CF M = job->fsrc[s].M, b = job->fsrc[s].b;
        const float *img = a.images + (long long)a.src_view[job->slot * a.n_src + s] * a.img_stride;
        float sr = 0.f, sv = 0.f, srr = 0.f, svv = 0.f, srv = 0.f;
        int cnt = 0;
        bool ok = true;
        for (int dy = -half; dy <= half && ok; dy += a.stride)
            for (int dx = -half; dx <= half; dx += a.stride) {
                const int qx = x + dx, qy = y + dy;
                if ((unsigned)qx >= (unsigned)W || (unsigned)qy >= (unsigned)H) { ok = false; break; }
                const float fqx = (float)qx, fqy = (float)qy;
                const float rqx = __builtin_fmaf(k1, fqy, __builtin_fmaf(k0, fqx, k2));
                const float rqy = __builtin_fmaf(k4, fqy, __builtin_fmaf(k3, fqx, k5));
                const float ndr = __builtin_fmaf(nx, rqx, __builtin_fmaf(ny, rqy, nz));
                const float t = delta / ndr;                  // depth of the plane along q's ray
                if (!(ndr < -1e-6f) || !(t > 0.0f)) { ok = false; break; }
                const float q0 = __builtin_fmaf(M[1], fqy, __builtin_fmaf(M[0], fqx, M[2]));
                const float q1 = __builtin_fmaf(M[4], fqy, __builtin_fmaf(M[3], fqx, M[5]));
                const float q2 = __builtin_fmaf(M[7], fqy, __builtin_fmaf(M[6], fqx, M[8]));
                const float p2 = __builtin_fmaf(t, q2, b[2]);
                if (!(p2 > 0.1f)) { ok = false; break; }
                const float rz = 1.0f / p2;
                const float u = __builtin_fmaf(t, q0, b[0]) * rz, v = __builtin_fmaf(t, q1, b[1]) * rz;
                bool inside;
                const float sval = xbilinear(img, H, W, u, v, inside);
                if (!inside) { ok = false; break; }
                const float rval = ref[AMVS_IDX((long long)qy * W + qx, (long long)H * W)];
                sr += rval; sv += sval; srr = __builtin_fmaf(rval, rval, srr); svv = __builtin_fmaf(sval, sval, svv);
                srv = __builtin_fmaf(rval, sval, srv);
                ++cnt;
            }
        if (!ok || cnt < 4) continue;
        const float inv = 1.0f / (float)cnt;
        const float cov = srv - sr * sv * inv, vr = srr - sr * sr * inv, vs = svv - sv * sv * inv;
        const float den = vr * vs;
        const float ncc = den > 1e-12f ? cov / __builtin_sqrtf(den) : 0.0f;
        xsorted_insert(costs, 1.0f - ncc);
        ++n_valid;
    }
    return xbetter_half(costs, n_valid);
}

// ---- the same cost for the common window shapes, N x N taps (patch = (N-1) stride + 1) ----
// Per pixel, once: the N*N reference taps and their sums.  Per hypothesis and source: the plane-induced
// homography  [uz', vz', z'] = (delta M + b (n^T K^-1)) q  (the common factor 1 / n.K^-1 q of the
// lifted point cancels in u = uz'/z'), advanced incrementally along a window row.  Every validity
// condition of xcost -- the plane faces the camera along the tap's ray, the lifted point is in front of
// the source, the footprint lies inside the source image -- is affine or projective in q, so it holds
// on the whole window iff it holds at its four corners: the taps run without tests.
template <int N> struct XRef {
    float r[N * N];
    float sr, srr;
    bool inside;
};

template <int N>
AMVS_DEV XRef<N> xref_load(const XArgs &a, const float *ref, int x, int y)
{
    XRef<N> R;
    const int half = a.patch / 2, st = a.stride;
    R.inside = (x - half >= 0) & (y - half >= 0) & (x - half + (N - 1) * st < a.W) & (y - half + (N - 1) * st < a.H);
    R.sr = 0.f; R.srr = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float v = R.inside ? ref[AMVS_IDX((long long)(y - half + j * st) * a.W + (x - half + i * st), (long long)a.H * a.W)] : 0.0f;
            R.r[j * N + i] = v;
            R.sr += v;
            R.srr = __builtin_fmaf(v, v, R.srr);
        }
    return R;
}

// U8: the sources are sampled from the packed 8-bit row-pair maps (one dword per 2 x 2 footprint,
// amvs_device.h) instead of four float loads; the sums run in code units and are rescaled once.
template <int N, bool U8>
AMVS_DEV float xcost_t(const XArgs &a, JobCP job, const XRef<N> &R, int x, int y, float rpx, float rpy, float d,
                       float nx, float ny, float nz)
{
    if (!R.inside) return __builtin_inff();
    const float ndr_p = __builtin_fmaf(nx, rpx, __builtin_fmaf(ny, rpy, nz));
    if (!(ndr_p < -1e-6f)) return __builtin_inff();           // plane must face the camera
    const float delta = d * ndr_p;                            // plane: n.X = delta
    const float k0 = job->Kinv[0], k1 = job->Kinv[1], k2 = job->Kinv[2], k3 = job->Kinv[3], k4 = job->Kinv[4],
                k5 = job->Kinv[5];
    // n . K^-1 q = w . [qx, qy, 1]
    const float w0 = __builtin_fmaf(ny, k3, nx * k0), w1 = __builtin_fmaf(ny, k4, nx * k1);
    const float w2 = __builtin_fmaf(ny, k5, __builtin_fmaf(nx, k2, nz));
    const float st = (float)a.stride;
    const float x0 = (float)(x - a.patch / 2), y0 = (float)(y - a.patch / 2);
    const float cx[2] = {x0, x0 + (float)(N - 1) * st}, cy[2] = {y0, y0 + (float)(N - 1) * st};
    float ndr_c[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ndr_c[c] = __builtin_fmaf(w0, cx[c & 1], __builtin_fmaf(w1, cy[c >> 1], w2));
    if (!(__builtin_fmaxf(__builtin_fmaxf(ndr_c[0], ndr_c[1]), __builtin_fmaxf(ndr_c[2], ndr_c[3])) < -1e-6f))
        return __builtin_inff();                               // (the generic path skips every source then)
    const float fw = (float)(a.W - 1), fh = (float)(a.H - 1);
    constexpr float INV = 1.0f / (float)(N * N);
    const float vr = R.srr - R.sr * R.sr * INV;
    float costs[AMVS_KMAX_SRC];
#pragma unroll
    for (int j = 0; j < AMVS_KMAX_SRC; ++j) costs[j] = __builtin_inff();
    int n_valid = 0;
    for (int s = 0; s < a.n_src; ++s) {
        const CF M = job->fsrc[s].M, b = job->fsrc[s].b;
        const int sview = a.src_view[job->slot * a.n_src + s];
        const float *img = a.images + (long long)sview * a.img_stride;
        const unsigned long long pimg = (unsigned long long)(a.pairs + (U8 ? (long long)sview * a.pair_stride : 0ll));
        constexpr int PB = AMVS_PAIR_BORDER;
        const int ppitch = a.W + 2 * PB;
        float Hm[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Hm[3 * r] = __builtin_fmaf(b[r], w0, delta * M[3 * r]);
            Hm[3 * r + 1] = __builtin_fmaf(b[r], w1, delta * M[3 * r + 1]);
            Hm[3 * r + 2] = __builtin_fmaf(b[r], w2, delta * M[3 * r + 2]);
        }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float qx = cx[c & 1], qy = cy[c >> 1];
            const float p0 = __builtin_fmaf(Hm[0], qx, __builtin_fmaf(Hm[1], qy, Hm[2]));
            const float p1 = __builtin_fmaf(Hm[3], qx, __builtin_fmaf(Hm[4], qy, Hm[5]));
            const float p2 = __builtin_fmaf(Hm[6], qx, __builtin_fmaf(Hm[7], qy, Hm[8]));
            // depth in the source p2 / ndr > 0.1 with ndr < 0
            ok &= p2 < 0.1f * ndr_c[c];
            const float rz = 1.0f / p2;
            const float u = p0 * rz, v = p1 * rz;
            ok &= (u >= 0.0f) & (v >= 0.0f) & (u < fw) & (v < fh);
        }
        if (!ok) continue;
        float sv = 0.f, svv = 0.f, srv = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float qy = y0 + (float)j * st;
            float p0 = __builtin_fmaf(Hm[0], x0, __builtin_fmaf(Hm[1], qy, Hm[2]));
            float p1 = __builtin_fmaf(Hm[3], x0, __builtin_fmaf(Hm[4], qy, Hm[5]));
            float p2 = __builtin_fmaf(Hm[6], x0, __builtin_fmaf(Hm[7], qy, Hm[8]));
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float rz = __builtin_amdgcn_rcpf(p2);
                const float u = p0 * rz, v = p1 * rz;
                const float x0f = __builtin_floorf(u), y0f = __builtin_floorf(v);
                // corner test + convexity: 0 <= x0f <= W-2 up to the rounding of the incremental chain, i.e.
                // -1 .. W-1 at worst.  The packed maps carry a zero border of AMVS_PAIR_BORDER = 2 texels, so
                // that range needs no clamp there (a footprint one texel outside reads zeros with a weight
                // of ~1e-7); the float maps have no border and keep the clamp.
                const int xi = U8 ? (int)x0f : min(max((int)x0f, 0), a.W - 2);
                const int yi = U8 ? (int)y0f : min(max((int)y0f, 0), a.H - 2);
                const float fx = u - x0f, fy = v - y0f;
                float top, bot;
                if constexpr (U8) {
                    const uint32_t wd = load_pair_word(pimg, AMVS_IDX((yi + PB) * ppitch + xi + PB, (a.H + 2 * PB) * ppitch - 1), 0);
                    // bytes: (y,x) (y+1,x) (y,x+1) (y+1,x+1)
                    const float t00 = (float)(wd & 0xFFu), t10 = (float)((wd >> 8) & 0xFFu);
                    const float t01 = (float)((wd >> 16) & 0xFFu), t11 = (float)(wd >> 24);
                    top = __builtin_fmaf(fx, t01 - t00, t00);
                    bot = __builtin_fmaf(fx, t11 - t10, t10);
                } else {
                    const float *pp = img + AMVS_IDX((long long)yi * a.W + xi, (long long)a.H * a.W - a.W - 1);
                    top = __builtin_fmaf(fx, pp[1] - pp[0], pp[0]);
                    bot = __builtin_fmaf(fx, pp[a.W + 1] - pp[a.W], pp[a.W]);
                }
                const float sval = __builtin_fmaf(fy, bot - top, top);
                sv += sval;
                svv = __builtin_fmaf(sval, sval, svv);
                srv = __builtin_fmaf(R.r[j * N + i], sval, srv);
                p0 = __builtin_fmaf(st, Hm[0], p0);
                p1 = __builtin_fmaf(st, Hm[3], p1);
                p2 = __builtin_fmaf(st, Hm[6], p2);
            }
        }
        if constexpr (U8) { sv *= (1.0f / 255.0f); srv *= (1.0f / 255.0f); svv *= (1.0f / 65025.0f); }
        const float cov = srv - R.sr * sv * INV, vs = svv - sv * sv * INV;
        const float den = vr * vs;
        const float ncc = den > 1e-12f ? cov * __builtin_amdgcn_rsqf(den) : 0.0f;
        xsorted_insert(costs, 1.0f - ncc);
        ++n_valid;
    }
    return xbetter_half(costs, n_valid);
}

AMVS_DEV void xnormalise_facing(float &nx, float &ny, float &nz)
{
    const float l = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
    const float il = l > 1e-12f ? 1.0f / l : 0.0f;
    nx *= il; ny *= il; nz *= il;
    if (!(nz < -0.05f)) { nx = 0.f; ny = 0.f; nz = -1.f; }    // keep the plane facing the camera
}

// random initialisation: log-uniform depth, normals around the viewing direction (mvs_patchmatch.py:268-284)
__global__ __launch_bounds__(256) void xpm_init_kernel(const XArgs a, float log_scale, float log_min)
{
    const JobCP job = (JobCP)(a.jobs + blockIdx.y);
    const long long HW = (long long)a.H * a.W;
    const StreamKey key = stream_key(a.seed, job->stream_view, 0u);
    const long long base = (long long)job->ref_img * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t h0 = pixel_hash((uint32_t)i, key);
        a.depth[base + i] = exp_poly(rng_uniform(h0) * log_scale + log_min);
        float g0, g1, g2;
        rng_normals3(h0, g0, g1, g2);
        float nx = g0 * 0.3f, ny = g1 * 0.3f, nz = -1.0f;
        xnormalise_facing(nx, ny, nz);
        a.normal[3 * (base + i)] = nx; a.normal[3 * (base + i) + 1] = ny; a.normal[3 * (base + i) + 2] = nz;
        a.cost[base + i] = __builtin_inff();
    }
}

// view propagation candidates from a snapshot of the maps: source a.colour (re-used as the source
// index of this iteration) of every job
__global__ __launch_bounds__(256) void xpm_view_candidates_kernel(const XArgs a)
{
    const JobCP job = (JobCP)(a.jobs + blockIdx.y);
    const int H = a.H, W = a.W, s = a.colour;
    const long long HW = (long long)H * W;
    const long long rbase = (long long)job->ref_img * HW, cbase = (long long)job->slot * HW;
    const int sv = a.src_view[job->slot * a.n_src + s];
    const float *sd = a.snap_depth + (long long)sv * HW, *sn = a.snap_normal + 3ll * sv * HW;
    const CF M = job->fsrc[s].M, b = job->fsrc[s].b;
    const CF Rr = job->Rref, tr = job->tref, Rs = job->src[s].R, ts = job->src[s].t, Ki = job->Kinv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        float cd = 0.0f, cnx = 0.f, cny = 0.f, cnz = -1.f;          // depth 0 = no candidate
        const float d = a.snap_depth[rbase + i];
        const float fx = (float)x, fy = (float)y;
        const float q0 = __builtin_fmaf(M[1], fy, __builtin_fmaf(M[0], fx, M[2]));
        const float q1 = __builtin_fmaf(M[4], fy, __builtin_fmaf(M[3], fx, M[5]));
        const float q2 = __builtin_fmaf(M[7], fy, __builtin_fmaf(M[6], fx, M[8]));
        const float p2 = __builtin_fmaf(d, q2, b[2]);
        if (p2 > 0.1f) {
            const int px = (int)__builtin_rintf(__builtin_fmaf(d, q0, b[0]) / p2);
            const int py = (int)__builtin_rintf(__builtin_fmaf(d, q1, b[1]) / p2);
            if ((unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H) {
                const long long j = AMVS_IDX((long long)py * W + px, HW);
                const float d2 = sd[j];
                const float n0 = sn[3 * j], n1 = sn[3 * j + 1], n2 = sn[3 * j + 2];
                // plane in the source frame: n'.Y = n'.(d' K^-1 p'), to the world, to the reference frame
                const float r0 = __builtin_fmaf(Ki[1], (float)py, __builtin_fmaf(Ki[0], (float)px, Ki[2]));
                const float r1 = __builtin_fmaf(Ki[4], (float)py, __builtin_fmaf(Ki[3], (float)px, Ki[5]));
                const float dl = d2 * (n0 * r0 + n1 * r1 + n2);
                float nw[3], nr[3];
                for (int c = 0; c < 3; ++c) nw[c] = Rs[c] * n0 + Rs[3 + c] * n1 + Rs[6 + c] * n2;       // R_s^T n'
                const float dw = dl - (n0 * ts[0] + n1 * ts[1] + n2 * ts[2]);
                for (int c = 0; c < 3; ++c) nr[c] = Rr[3 * c] * nw[0] + Rr[3 * c + 1] * nw[1] + Rr[3 * c + 2] * nw[2];   // R_r n_w
                const float drf = dw + (nr[0] * tr[0] + nr[1] * tr[1] + nr[2] * tr[2]);
                const float rpx = __builtin_fmaf(Ki[1], fy, __builtin_fmaf(Ki[0], fx, Ki[2]));
                const float rpy = __builtin_fmaf(Ki[4], fy, __builtin_fmaf(Ki[3], fx, Ki[5]));
                const float ndr = nr[0] * rpx + nr[1] * rpy + nr[2];
                const float t = drf / ndr;
                if (ndr < -1e-6f && t >= a.depth_min && t <= a.depth_max) {
                    cd = t; cnx = nr[0]; cny = nr[1]; cnz = nr[2];
                    xnormalise_facing(cnx, cny, cnz);
                }
            }
        }
        (void)AMVS_IDX(job->slot, a.n_jobs);                    // (the candidates are indexed by the job's slot)
        a.cand_d[cbase + i] = cd;
        a.cand_n[3 * (cbase + i)] = cnx; a.cand_n[3 * (cbase + i) + 1] = cny; a.cand_n[3 * (cbase + i) + 2] = cnz;
    }
}

// one red-black half sweep: spatial propagation from the four neighbours of the other colour, the
// view candidate, refinement; in place
template <int NT, bool U8>
__global__ __launch_bounds__(128) void xpm_sweep_kernel(const XArgs a)
{
    const JobCP job = (JobCP)(a.jobs + blockIdx.y);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const long long rbase = (long long)job->ref_img * HW, cbase = (long long)job->slot * HW;
    const float *ref = a.images + (long long)job->ref_img * a.img_stride;
    float *D = a.depth + rbase, *N = a.normal + 3 * rbase, *C = a.cost + rbase;
    const CF Ki = job->Kinv;
    const int hw = (W + 1) / 2;                         // pixels of one colour per row (at most)
    const long long half_n = (long long)H * hw;
    for (long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x; h < half_n; h += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(h / hw);
        const int x = 2 * (int)(h - (long long)y * hw) + ((y + a.colour) & 1);
        if (x >= W) continue;
        const long long i = AMVS_IDX((long long)y * W + x, HW);
        float bd = D[i], bnx = N[3 * i], bny = N[3 * i + 1], bnz = N[3 * i + 2];
        float bc = C[i];
        const float rpx = __builtin_fmaf(Ki[1], (float)y, __builtin_fmaf(Ki[0], (float)x, Ki[2]));
        const float rpy = __builtin_fmaf(Ki[4], (float)y, __builtin_fmaf(Ki[3], (float)x, Ki[5]));
        constexpr int NR = NT > 0 ? NT : 1;
        const XRef<NR> R = NT > 0 ? xref_load<NR>(a, ref, x, y) : XRef<NR>{};
        // Hypotheses in order: 0 the current plane (only while its cost is unknown), 1-4 the planes of the
        // four neighbours of the other colour extended to this pixel's ray, 5 the view candidate,
        // 6.. the refinements of the best so far, last a fresh random plane.  ONE call site of the cost
        // (the window loops are unrolled: inlining it per hypothesis cost 230 VGPRs and scratch memory).
        const int n_hyp = 6 + a.n_refine + (a.with_random ? 1 : 0);
        for (int hyp = 0; hyp < n_hyp; ++hyp) {
            float d = bd, nx = bnx, ny = bny, nz = bnz;
            bool have = true;
            if (hyp == 0) {
                have = !(bc < __builtin_inff());
            } else if (hyp <= 4) {
                const int k = hyp - 1;
                const int xx = x + (k == 0 ? -1 : (k == 1 ? 1 : 0)), yy = y + (k == 2 ? -1 : (k == 3 ? 1 : 0));
                have = ((unsigned)xx < (unsigned)W) & ((unsigned)yy < (unsigned)H);
                if (have) {
                    const long long j = AMVS_IDX((long long)yy * W + xx, HW);
                    const float nd = D[j];
                    nx = N[3 * j]; ny = N[3 * j + 1]; nz = N[3 * j + 2];
                    const float rqx = __builtin_fmaf(Ki[1], (float)yy, __builtin_fmaf(Ki[0], (float)xx, Ki[2]));
                    const float rqy = __builtin_fmaf(Ki[4], (float)yy, __builtin_fmaf(Ki[3], (float)xx, Ki[5]));
                    const float dl = nd * (nx * rqx + ny * rqy + nz);
                    const float ndr = nx * rpx + ny * rpy + nz;
                    have = ndr < -1e-6f;
                    d = dl / ndr;
                }
            } else if (hyp == 5) {
                have = a.with_view_cand != 0;
                if (have) {
                    d = a.cand_d[cbase + i];
                    have = d > 0.0f;
                    nx = a.cand_n[3 * (cbase + i)]; ny = a.cand_n[3 * (cbase + i) + 1]; nz = a.cand_n[3 * (cbase + i) + 2];
                }
            } else if (hyp < 6 + a.n_refine) {
                const int r = hyp - 6;
                const StreamKey key = stream_key(a.seed, job->stream_view, a.draw * 8u + (unsigned)r);
                const uint32_t h0 = pixel_hash((uint32_t)i, key);
                float g0, g1, g2;
                rng_normals3(h0, g0, g1, g2);
                const float scale = r == 0 ? 1.0f : 0.25f;          // a wide and a narrow perturbation
                d = bd * (1.0f + (rng_uniform(h0) * 2.0f - 1.0f) * a.rel_range * scale);
                nx = bnx + g0 * a.nrm_range * scale; ny = bny + g1 * a.nrm_range * scale; nz = bnz + g2 * a.nrm_range * scale;
                xnormalise_facing(nx, ny, nz);
            } else {
                const StreamKey key = stream_key(a.seed, job->stream_view, a.draw * 8u + 7u);
                const uint32_t h0 = pixel_hash((uint32_t)i, key);
                float g0, g1, g2;
                rng_normals3(h0, g0, g1, g2);
                const float lmin = __builtin_logf(a.depth_min), lmax = __builtin_logf(a.depth_max);
                nx = g0 * 0.3f; ny = g1 * 0.3f; nz = -1.0f;
                xnormalise_facing(nx, ny, nz);
                d = __builtin_expf(lmin + rng_uniform(h0) * (lmax - lmin));
            }
            if (hyp > 0) have = have & (d >= a.depth_min) & (d <= a.depth_max);
            if (!have) continue;
            float c;
            if constexpr (NT > 0) c = xcost_t<NT, U8>(a, job, R, x, y, rpx, rpy, d, nx, ny, nz);
            else c = xcost(a, job, ref, x, y, d, nx, ny, nz);
            if (hyp == 0 || c < bc) { bc = c; bd = d; bnx = nx; bny = ny; bnz = nz; }
        }
        D[i] = bd; N[3 * i] = bnx; N[3 * i + 1] = bny; N[3 * i + 2] = bnz; C[i] = bc;
    }
}

// test hook (amvs_xpm_step, phase 3): the cost of every pixel's current plane, no selection
template <int NT, bool U8>
__global__ __launch_bounds__(128) void xpm_eval_kernel(const XArgs a, float *__restrict__ cost_out)
{
    const JobCP job = (JobCP)(a.jobs + blockIdx.y);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const long long rbase = (long long)job->ref_img * HW, cbase = (long long)job->slot * HW;
    const float *ref = a.images + (long long)job->ref_img * a.img_stride;
    const CF Ki = job->Kinv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const float d = a.depth[rbase + i];
        const float nx = a.normal[3 * (rbase + i)], ny = a.normal[3 * (rbase + i) + 1], nz = a.normal[3 * (rbase + i) + 2];
        const float rpx = __builtin_fmaf(Ki[1], (float)y, __builtin_fmaf(Ki[0], (float)x, Ki[2]));
        const float rpy = __builtin_fmaf(Ki[4], (float)y, __builtin_fmaf(Ki[3], (float)x, Ki[5]));
        constexpr int NR = NT > 0 ? NT : 1;
        const XRef<NR> R = NT > 0 ? xref_load<NR>(a, ref, x, y) : XRef<NR>{};
        float c;
        if constexpr (NT > 0) c = xcost_t<NT, U8>(a, job, R, x, y, rpx, rpy, d, nx, ny, nz);
        else c = xcost(a, job, ref, x, y, d, nx, ny, nz);
        cost_out[cbase + i] = c;
    }
}

// geometric consistency: sources whose own map agrees after forward-backward reprojection
__global__ __launch_bounds__(256) void xpm_consistency_kernel(const XArgs a, float *__restrict__ conf_out, float max_px,
                                                              float max_rel)
{
    const JobCP job = (JobCP)(a.jobs + blockIdx.y);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const long long rbase = (long long)job->ref_img * HW, cbase = (long long)job->slot * HW;
    const CF Rr = job->Rref, tr = job->tref, Ki = job->Kinv, K = job->K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const float d = a.depth[rbase + i];
        float cnt = 0.0f;
        if (a.cost[rbase + i] < 0.6f) {                               // photometrically plausible at all
            for (int s = 0; s < a.n_src; ++s) {
                const CF M = job->fsrc[s].M, b = job->fsrc[s].b, Rs = job->src[s].R, ts = job->src[s].t;
                const int sv = a.src_view[job->slot * a.n_src + s];
                const float fx = (float)x, fy = (float)y;
                const float q0 = __builtin_fmaf(M[1], fy, __builtin_fmaf(M[0], fx, M[2]));
                const float q1 = __builtin_fmaf(M[4], fy, __builtin_fmaf(M[3], fx, M[5]));
                const float q2 = __builtin_fmaf(M[7], fy, __builtin_fmaf(M[6], fx, M[8]));
                const float p2 = __builtin_fmaf(d, q2, b[2]);
                if (!(p2 > 0.1f)) continue;
                const int px = (int)__builtin_rintf(__builtin_fmaf(d, q0, b[0]) / p2);
                const int py = (int)__builtin_rintf(__builtin_fmaf(d, q1, b[1]) / p2);
                if ((unsigned)px >= (unsigned)W || (unsigned)py >= (unsigned)H) continue;
                const float d2 = a.depth[(long long)sv * HW + AMVS_IDX((long long)py * W + px, HW)];
                // the source's point -> world -> reference camera -> pixel
                float Y[3] = {__builtin_fmaf(Ki[1], (float)py, __builtin_fmaf(Ki[0], (float)px, Ki[2])) * d2,
                              __builtin_fmaf(Ki[4], (float)py, __builtin_fmaf(Ki[3], (float)px, Ki[5])) * d2, d2};
                float Xw[3], Xr[3];
                for (int c = 0; c < 3; ++c)
                    Xw[c] = Rs[c] * (Y[0] - ts[0]) + Rs[3 + c] * (Y[1] - ts[1]) + Rs[6 + c] * (Y[2] - ts[2]);
                for (int c = 0; c < 3; ++c) Xr[c] = Rr[3 * c] * Xw[0] + Rr[3 * c + 1] * Xw[1] + Rr[3 * c + 2] * Xw[2] + tr[c];
                if (!(Xr[2] > 0.1f)) continue;
                const float u = (K[0] * Xr[0] + K[1] * Xr[1]) / Xr[2] + K[2], v = (K[3] * Xr[0] + K[4] * Xr[1]) / Xr[2] + K[5];
                const float eu = u - fx, ev = v - fy;
                if (eu * eu + ev * ev < max_px * max_px && __builtin_fabsf(Xr[2] - d) < max_rel * d) cnt += 1.0f;
            }
        }
        conf_out[cbase + i] = cnt;
    }
}

static dim3 xgrid(long long n, int tpb, int n_jobs)
{
    long long b = (n + tpb - 1) / tpb;
    return dim3((unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)), (unsigned)n_jobs);
}

hipError_t launch_xpm_init(const XArgs &a, float log_scale, float log_min, hipStream_t st)
{
    hipLaunchKernelGGL(xpm_init_kernel, xgrid((long long)a.H * a.W, 256, a.n_jobs), dim3(256), 0, st, a, log_scale, log_min);
    return hipGetLastError();
}

hipError_t launch_xpm_view_candidates(const XArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(xpm_view_candidates_kernel, xgrid((long long)a.H * a.W, 256, a.n_jobs), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_xpm_sweep(const XArgs &a, hipStream_t st)
{
    const dim3 grid = xgrid(((long long)a.H * a.W + 1) / 2, 128, a.n_jobs), blk(128);
    // taps per axis; the specialised cost needs patch = (N - 1) stride + 1 with 3 <= N <= 7
    const int n = (a.patch - 1) / a.stride + 1;
    const bool fits = (n - 1) * a.stride + 1 == a.patch;
    const bool u8 = a.pairs != nullptr;
#define AMVS_XSWEEP(NT_)                                                                                      \
    if (u8) hipLaunchKernelGGL((xpm_sweep_kernel<NT_, true>), grid, blk, 0, st, a);                              \
    else hipLaunchKernelGGL((xpm_sweep_kernel<NT_, false>), grid, blk, 0, st, a)
    switch (fits ? n : 0) {
    case 3: AMVS_XSWEEP(3); break;
    case 4: AMVS_XSWEEP(4); break;
    case 5: AMVS_XSWEEP(5); break;
    case 6: AMVS_XSWEEP(6); break;
    case 7: AMVS_XSWEEP(7); break;
    default: hipLaunchKernelGGL((xpm_sweep_kernel<0, false>), grid, blk, 0, st, a); break;
    }
#undef AMVS_XSWEEP
    return hipGetLastError();
}

hipError_t launch_xpm_eval(const XArgs &a, float *cost_out, hipStream_t st)
{
    const dim3 grid = xgrid((long long)a.H * a.W, 128, a.n_jobs), blk(128);
    const int n = (a.patch - 1) / a.stride + 1;
    const bool fits = (n - 1) * a.stride + 1 == a.patch;
    const bool u8 = a.pairs != nullptr;
#define AMVS_XEVAL(NT_)                                                                                       \
    if (u8) hipLaunchKernelGGL((xpm_eval_kernel<NT_, true>), grid, blk, 0, st, a, cost_out);                     \
    else hipLaunchKernelGGL((xpm_eval_kernel<NT_, false>), grid, blk, 0, st, a, cost_out)
    switch (fits ? n : 0) {
    case 3: AMVS_XEVAL(3); break;
    case 4: AMVS_XEVAL(4); break;
    case 5: AMVS_XEVAL(5); break;
    case 6: AMVS_XEVAL(6); break;
    case 7: AMVS_XEVAL(7); break;
    default: hipLaunchKernelGGL((xpm_eval_kernel<0, false>), grid, blk, 0, st, a, cost_out); break;
    }
#undef AMVS_XEVAL
    return hipGetLastError();
}

hipError_t launch_xpm_consistency(const XArgs &a, float *conf_out, float max_px, float max_rel, hipStream_t st)
{
    hipLaunchKernelGGL(xpm_consistency_kernel, xgrid((long long)a.H * a.W, 256, a.n_jobs), dim3(256), 0, st, a, conf_out,
                       max_px, max_rel);
    return hipGetLastError();
}

}  // namespace amvs

AMVS_CHECK_TU(extended)
