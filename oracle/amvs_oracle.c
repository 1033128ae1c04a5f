/*
 * amvs_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the dense-reconstruction hot path of
 * dackey-wav/3d-reconstruction-tool:
 *     src/core/mvs_patchmatch.py   (PatchMatchMVS._patchmatch_cuda and below)
 *     src/core/dense_stereo.py     (DenseStereoReconstructor._plane_sweep_torch)
 * Every function cites the reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file;
 * the product path (3d-reconstruction-tool_amd/) never links or calls it.
 *
 * PARITY PINNING: the reference has no tests or golden vectors of its own
 * (SURVEY.md section 4).  This oracle is pinned against outputs of the reference itself,
 * captured in the build container by tests/golden/make_golden.py and committed
 * as tests/golden/g*.npz ; tests/test_oracle_golden.py checks every fixture.
 *
 * Arithmetic contract (all float32, IEEE, no contraction except explicit fmaf):
 *   - 3-term contractions follow torch-CPU matmul order, measured bit-exact in
 *     the build container:  fma(a2,b2, fma(a1,b1, a0*b0)).
 *   - bilinear sampling follows ATen's vectorised grid_sample (align_corners,
 *     zeros padding), measured bit-exact: fma(se,t11, fma(sw,t10, fma(ne,t01, nw*t00))).
 *   - the k x k box filter (reference: F.conv2d with a ones/k^2 kernel, whose
 *     summation order inside oneDNN is not observable) uses the order the HIP
 *     kernels use: column sums top->bottom (plain sum for v, fma chains for
 *     v*v and r*v), then row sums right->left, then one multiply by 1/k^2.
 *     This differs from the reference by summation order only (tolerance in tests).
 *   - RNG: the reference never seeds (torch.rand / torch.randn); "identical RNG
 *     streams" are defined by the counter-hash generator below, which the golden
 *     capture injects into the reference in place of torch.rand/randn.
 *
 * TWO ARITHMETIC MODES (orc_ctx_set_mode):
 *   exact (0): the contract above -- what the reference computes, operation for operation
 *              up to the box-filter summation order.
 *   fast  (1): the "tolerance mode" of the HIP backend (include/amvs.h AMVS_MODE_FAST), restated
 *              here so that HIP-fast can be checked BIT FOR BIT against this file while this
 *              file is checked against the reference's golden vectors within the tolerances
 *              tests/test_oracle_golden.py states.  Same algorithm, cheaper arithmetic:
 *                - images are 8-bit codes (gray = code/255 exactly); sums run in code units;
 *                - projection precomposed per source: [u z, v z, z] = d * (M [x,y,1]) + b with
 *                  M = K R_s R_ref^T K^-1, b = K (t_s - R_s R_ref^T t_ref) formed in double;
 *                - one reciprocal of z, no Markstein quotient refinement, no
 *                  normalise / un-normalise round trip around grid_sample;
 *                - bilinear sample as two horizontal lerps and one vertical lerp;
 *                - reference-image window sums are exact integers.
 *              Everything else (RNG, candidates, select, NaN/inf conventions) is shared.
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))
#define ORC_MAX_SRC 16

/* ------------------------------------------------------------------ RNG -- */
/* Counter-hash generator.  One "draw" = the random numbers one pixel consumes
 * in one reference step:
 *   draw 0            : init  -> rand(H,W), randn(H,W), randn(H,W)   (mvs_patchmatch.py:271,279,280)
 *   draw 1+it*ns+s    : refinement sample s of iteration it
 *                       -> rand(H,W), randn(H,W,3)                   (mvs_patchmatch.py:471,475)
 * Per pixel a draw yields one uniform U in [0,1) and normals N0,N1,N2.       */

static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

static inline void stream_keys(uint64_t seed, uint32_t view, uint32_t draw,
                               uint32_t *k1, uint32_t *k2)
{
    uint32_t a = fmix32((uint32_t)seed ^ 0x9E3779B9u);
    a = fmix32(a + view);
    uint32_t b = fmix32((uint32_t)(seed >> 32) ^ 0x85EBCA6Bu);
    b = fmix32(b + draw);
    b = fmix32(b ^ a);
    *k1 = a; *k2 = b;
}

static inline uint32_t pixel_hash(uint32_t idx, uint32_t k1, uint32_t k2)
{
    return fmix32(fmix32(idx ^ k1) + k2);
}

static inline float rng_uniform(uint32_t h0)
{
    return (float)(h0 >> 8) * 0x1p-24f;
}

/* natural log for t in [0.5, 65536): exponent split + degree-9 polynomial. */
static inline float orc_logf(float t)
{
    union { float f; uint32_t u; } c; c.f = t;
    int e = (int)((c.u >> 23) & 0xFF) - 127;
    c.u = (c.u & 0x007FFFFFu) | 0x3F800000u;     /* m in [1,2) */
    float m = c.f;
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* sin and cos of th in [0, pi/2] by Taylor polynomials in th^2. */
static inline void orc_sincos_q(float th, float *s, float *c)
{
    float t2 = th * th;
    float ps = 1.6059043837e-10f;            /* 1/13! */
    ps = fmaf(ps, t2, -2.5052108385e-8f);    /* -1/11! */
    ps = fmaf(ps, t2, 2.7557319224e-6f);     /* 1/9!  */
    ps = fmaf(ps, t2, -1.9841269841e-4f);    /* -1/7! */
    ps = fmaf(ps, t2, 8.3333333333e-3f);     /* 1/5!  */
    ps = fmaf(ps, t2, -1.6666666667e-1f);    /* -1/3! */
    *s = fmaf(ps * t2, th, th);
    float pc = 2.0876756988e-9f;             /* 1/12! */
    pc = fmaf(pc, t2, -2.7557319224e-7f);    /* -1/10! */
    pc = fmaf(pc, t2, 2.4801587302e-5f);     /* 1/8!  */
    pc = fmaf(pc, t2, -1.3888888889e-3f);    /* -1/6! */
    pc = fmaf(pc, t2, 4.1666666667e-2f);     /* 1/4!  */
    pc = fmaf(pc, t2, -0.5f);
    *c = fmaf(pc, t2, 1.0f);
}

/* Box-Muller pair from one 32-bit word (16-bit radius index, 16-bit angle). */
static inline void rng_normal_pair(uint32_t w, float *n0, float *n1)
{
    uint32_t a = w >> 16, b = w & 0xFFFFu;
    float t = (float)a + 0.5f;                               /* u1 = t / 65536 */
    float lnu = orc_logf(t) + (-11.090354888959125f);        /* - 16 ln 2     */
    float r = sqrtf(-2.0f * lnu);
    uint32_t q = b >> 14;
    float th = ((float)(b & 0x3FFFu) * 0x1p-14f) * 1.57079632679489662f;
    float s, c;
    orc_sincos_q(th, &s, &c);
    float cs, sn;
    switch (q) {
    case 0: cs = c;  sn = s;  break;
    case 1: cs = -s; sn = c;  break;
    case 2: cs = -c; sn = -s; break;
    default: cs = s; sn = -c; break;
    }
    *n0 = r * cs; *n1 = r * sn;
}

static inline void rng_draw(uint32_t idx, uint32_t k1, uint32_t k2,
                            float *u, float *n0, float *n1, float *n2)
{
    uint32_t h0 = pixel_hash(idx, k1, k2);
    *u = rng_uniform(h0);
    float spare;
    rng_normal_pair(fmix32(h0 + 0x9E3779B9u), n0, n1);
    rng_normal_pair(fmix32(h0 + 0x3C6EF372u), n2, &spare);
}

/* Fill U (n) and N (n x 3, interleaved) for one draw: the tensors the golden
 * capture injects into the reference in place of torch.rand / torch.randn. */
ORC_API void orc_rng_fill(uint64_t seed, uint32_t view, uint32_t draw, int64_t n,
                          float *u_out, float *n_out)
{
    uint32_t k1, k2;
    stream_keys(seed, view, draw, &k1, &k2);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float u, a, b, c;
        rng_draw((uint32_t)i, k1, k2, &u, &a, &b, &c);
        if (u_out) u_out[i] = u;
        if (n_out) { n_out[3 * i] = a; n_out[3 * i + 1] = b; n_out[3 * i + 2] = c; }
    }
}

/* exp for the log-uniform depth init (range-reduced degree-6 polynomial). */
static inline float orc_expf(float x)
{
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int ni = (int)n;
    if (ni > 127) ni = 127;
    if (ni < -126) ni = -126;
    union { float f; uint32_t u; } s; s.u = (uint32_t)(ni + 127) << 23;
    return y * s.f;
}

ORC_API float orc_expf_export(float x) { return orc_expf(x); }

/* -------------------------------------------------------------- camera -- */
typedef struct {
    int H, W;
    float K[9], Kinv[9];      /* row-major */
    float Rref[9], tref[3];   /* world->camera of the reference view */
} orc_cam_t;

/* a / b given rb = RN(1/b): product, one fma residual, one fma correction.  With a correctly
 * rounded reciprocal this is the correctly rounded quotient (Markstein); it matched IEEE
 * division on 2.16e8 random operand pairs and keeps the sampling stage bit-exact against
 * torch-CPU.  Written out (instead of `/`) because the HIP kernels share one reciprocal
 * among the quotients of a common denominator and the oracle follows their arithmetic. */
static inline float qdiv(float a, float b, float rb)
{
    float q = a * rb;
    float r = fmaf(-q, b, a);
    return fmaf(r, rb, q);
}

/* F.normalize(v, dim=-1): v / max(||v||_2, 1e-12)   (mvs_patchmatch.py:281,476) */
static inline void normalize3(float *x, float *y, float *z)
{
    float n = sqrtf((*x) * (*x) + (*y) * (*y) + (*z) * (*z));
    float d = n > 1e-12f ? n : 1e-12f;
    float rd = 1.0f / d;
    *x = qdiv(*x, d, rd); *y = qdiv(*y, d, rd); *z = qdiv(*z, d, rd);
}

/* Back-project pixel (x,y) at depth d to world coordinates.
 * mvs_patchmatch.py:341-347 (also :500-504, dense_stereo.py:267-271):
 *   rays = [x,y,1] @ K_inv.T ; X = rays*d ; Xw = (X - t_ref) @ R_ref        */
static inline void backproject(const orc_cam_t *c, int x, int y, float d, float Pw[3])
{
    float px = (float)x, py = (float)y;
    float q[3];
    for (int i = 0; i < 3; ++i) {
        float ray = fmaf(1.0f, c->Kinv[3 * i + 2],
                         fmaf(py, c->Kinv[3 * i + 1], px * c->Kinv[3 * i + 0]));
        q[i] = ray * d - c->tref[i];
    }
    for (int j = 0; j < 3; ++j)
        Pw[j] = fmaf(q[2], c->Rref[6 + j], fmaf(q[1], c->Rref[3 + j], q[0] * c->Rref[j]));
}

/* Project a world point into a source view and sample it bilinearly.
 * mvs_patchmatch.py:351-377 (projection, bounds, grid_sample with
 * align_corners=True / zeros padding).  bounds: 0 = patch bounds (:362-363),
 * 1 = image bounds (:516-517), 2 = depth test only (dense_stereo.py:280,303). */
static inline float project_sample(const orc_cam_t *c, const float Pw[3],
                                   const float *Rs, const float *ts,
                                   const float *img, int half, int bounds, int *valid)
{
    const int H = c->H, W = c->W;
    float ps[3];
    for (int i = 0; i < 3; ++i)
        ps[i] = fmaf(Pw[2], Rs[3 * i + 2], fmaf(Pw[1], Rs[3 * i + 1], Pw[0] * Rs[3 * i])) + ts[i];
    float z = ps[2];
    float zz = z + 1e-8f;
    float rz = 1.0f / zz;
    float a = qdiv(ps[0], zz, rz), b = qdiv(ps[1], zz, rz);
    float u = fmaf(b, c->K[1], a * c->K[0]) + c->K[2];
    float v = fmaf(b, c->K[4], a * c->K[3]) + c->K[5];
    int ok = z > 0.1f;
    if (bounds == 0)
        ok = ok && (u >= (float)half) && (u < (float)(W - half)) &&
             (v >= (float)half) && (v < (float)(H - half));
    else if (bounds == 1)
        ok = ok && (u >= 0.0f) && (u < (float)W) && (v >= 0.0f) && (v < (float)H);
    *valid = ok;
    /* proj_norm (:367-369) then ATen's unnormalise (g+1)*((size-1)/2) */
    float fw = (float)(W - 1), fh = (float)(H - 1);
    float gx = qdiv(2.0f * u, fw, 1.0f / fw) - 1.0f;
    float gy = qdiv(2.0f * v, fh, 1.0f / fh) - 1.0f;
    float ux = (gx + 1.0f) * (fw * 0.5f);
    float uy = (gy + 1.0f) * (fh * 0.5f);
    float x0 = floorf(ux), y0 = floorf(uy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float wx1 = ux - x0, wx0 = x1 - ux, wy1 = uy - y0, wy0 = y1 - uy;
    float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    int x0ok = (x0 >= 0.0f) && (x0 <= fw), x1ok = (x1 >= 0.0f) && (x1 <= fw);
    int y0ok = (y0 >= 0.0f) && (y0 <= fh), y1ok = (y1 >= 0.0f) && (y1 <= fh);
    float t00 = (x0ok && y0ok) ? img[(int)y0 * W + (int)x0] : 0.0f;
    float t01 = (x1ok && y0ok) ? img[(int)y0 * W + (int)x1] : 0.0f;
    float t10 = (x0ok && y1ok) ? img[(int)y1 * W + (int)x0] : 0.0f;
    float t11 = (x1ok && y1ok) ? img[(int)y1 * W + (int)x1] : 0.0f;
    return fmaf(t11, se, fmaf(t10, sw, fmaf(t01, ne, t00 * nw)));
}

/* ---------------------------------------------------------- box filter -- */
/* k x k box sums with zero padding (F.conv2d(..., padding=k//2) with a
 * ones/k^2 kernel: mvs_patchmatch.py:397-408, dense_stereo.py:325-341).
 * mode 0: plain sum of a; mode 1: sum of a*b via fma chain.               */
static void box_sum(const float *a, const float *b, int mode, int H, int W, int k,
                    float *tmp, float *out)
{
    const int h = k / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float acc = 0.0f;
            for (int i = -h; i <= h; ++i) {
                int yy = y + i;
                float av = (yy >= 0 && yy < H) ? a[yy * W + x] : 0.0f;
                float bv = (mode && yy >= 0 && yy < H) ? b[yy * W + x] : 0.0f;
                if (i == -h) acc = mode ? av * bv : av;
                else acc = mode ? fmaf(av, bv, acc) : acc + av;
            }
            tmp[y * W + x] = acc;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float acc = 0.0f;
            for (int j = h; j >= -h; --j) {
                int xx = x + j;
                float cv = (xx >= 0 && xx < W) ? tmp[y * W + xx] : 0.0f;
                acc = (j == h) ? cv : acc + cv;
            }
            out[y * W + x] = acc;
        }
    }
}

/* mean1 / var1 of an image (mvs_patchmatch.py:403,406 -- the reference
 * recomputes them every call; they depend on the image only).             */
ORC_API void orc_box_stats(const float *img, int H, int W, int k, float *mean, float *var)
{
    float *tmp = (float *)malloc(sizeof(float) * H * W);
    float *s2 = (float *)malloc(sizeof(float) * H * W);
    const float inv = 1.0f / (float)(k * k);
    box_sum(img, NULL, 0, H, W, k, tmp, mean);
    box_sum(img, img, 1, H, W, k, tmp, s2);
    for (int i = 0; i < H * W; ++i) {
        float m = mean[i] * inv;
        mean[i] = m;
        var[i] = s2[i] * inv - m * m;
    }
    free(tmp); free(s2);
}

/* NCC between img1 (with precomputed mean1/var1) and img2.
 * variant 0: cost = 1 - cov/(sqrt(var1*var2)+1e-8)        mvs_patchmatch.py:410-411
 * variant 1: ncc  = cov/sqrt(var1*var2+1e-8)              dense_stereo.py:344-345 */
static void ncc_map(const float *img1, const float *mean1, const float *var1,
                    const float *img2, int H, int W, int k, int variant,
                    float *tmp, float *b_v, float *b_vv, float *b_rv, float *out)
{
    const float inv = 1.0f / (float)(k * k);
    box_sum(img2, NULL, 0, H, W, k, tmp, b_v);
    box_sum(img2, img2, 1, H, W, k, tmp, b_vv);
    box_sum(img1, img2, 1, H, W, k, tmp, b_rv);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < H * W; ++i) {
        float mean2 = b_v[i] * inv;
        float var2 = b_vv[i] * inv - mean2 * mean2;
        float cov = b_rv[i] * inv - mean1[i] * mean2;
        /* quotients are formed as qdiv(a, b, RN(1/b)) (see qdiv above) */
        if (variant == 0) {
            float den = sqrtf(var1[i] * var2) + 1e-8f;
            float ncc = qdiv(cov, den, 1.0f / den);
            out[i] = 1.0f - ncc;
        } else {
            float den = sqrtf(var1[i] * var2 + 1e-8f);
            out[i] = qdiv(cov, den, 1.0f / den);
        }
    }
}

/* constants of the fast NCC: sums are in code units, statistics in gray units */
static inline float fast_c1(int k) { return (float)(1.0 / ((double)(k * k) * 255.0)); }
static inline float fast_c2(int k) { return (float)(1.0 / ((double)(k * k) * 65025.0)); }

/* Fast-mode NCC: img1c / img2 in code units, m1 / v1 the precomputed ref statistics in gray
 * units; the window sums keep the exact mode's order, the epilogue drops the Markstein
 * quotient refinement (one multiply by the reciprocal). */
static void ncc_map_fast(const float *img1c, const float *m1, const float *v1,
                         const float *img2, int H, int W, int k, int variant, float thresh2,
                         float *tmp, float *b_v, float *b_vv, float *b_rv, float *out)
{
    const float C1 = fast_c1(k), C2 = fast_c2(k);
    box_sum(img2, NULL, 0, H, W, k, tmp, b_v);
    box_sum(img2, img2, 1, H, W, k, tmp, b_vv);
    box_sum(img1c, img2, 1, H, W, k, tmp, b_rv);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < H * W; ++i) {
        float mean2 = b_v[i] * C1;
        float var2 = fmaf(-mean2, mean2, b_vv[i] * C2);
        float cov = fmaf(-m1[i], mean2, b_rv[i] * C2);
        if (variant == 0) {
            float den = sqrtf(v1[i] * var2) + 1e-8f;
            out[i] = 1.0f - cov * (1.0f / den);
        } else if (variant == 1) {
            float den = sqrtf(v1[i] * var2 + 1e-8f);
            out[i] = cov * (1.0f / den);
        } else {
            /* variant 2: the plane sweep's vote  ncc > thresh  (dense_stereo.py:303) for thresh > 0 without
             * the square root and the division:  cov / sqrt(x) > t  <=>  cov > 0, x >= 0 (a negative x is
             * the reference's NaN: no vote) and cov^2 > t^2 x.  out = 1 where the vote is cast. */
            float x = v1[i] * var2 + 1e-8f;
            out[i] = (cov > 0.0f && x >= 0.0f && cov * cov > thresh2 * x) ? 1.0f : 0.0f;
        }
    }
}

/* _ncc_cost (mvs_patchmatch.py:392-413) / _compute_ncc_torch (dense_stereo.py:318-347) */
ORC_API void orc_ncc(const float *img1, const float *img2, int H, int W, int k,
                     int variant, float *out)
{
    size_t n = (size_t)H * W;
    float *buf = (float *)malloc(sizeof(float) * n * 6);
    float *mean1 = buf, *var1 = buf + n, *tmp = buf + 2 * n, *bv = buf + 3 * n,
          *bvv = buf + 4 * n, *brv = buf + 5 * n;
    orc_box_stats(img1, H, W, k, mean1, var1);
    ncc_map(img1, mean1, var1, img2, H, W, k, variant, tmp, bv, bvv, brv, out);
    free(buf);
}

/* ------------------------------------------------------- view context -- */
typedef struct {
    orc_cam_t cam;
    int S, k;
    const float *ref;                 /* H*W */
    const float *src[ORC_MAX_SRC];    /* H*W each */
    float Rs[ORC_MAX_SRC][9], ts[ORC_MAX_SRC][3];
    float *mean1, *var1;              /* ref stats */
    float *sampled, *tmp, *bv, *bvv, *brv, *cost_s, *total, *count;
    unsigned char *valid;
    /* fast mode (orc_ctx_set_mode) */
    int mode;
    unsigned char *ref_code, *src_code[ORC_MAX_SRC];   /* 8-bit codes of the images */
    float *ref_codef;                                   /* the same as floats (0..255) */
    float M[ORC_MAX_SRC][9], bvec[ORC_MAX_SRC][3];      /* precomposed projections */
    float *m1f, *v1f;                                   /* ref mean / variance (gray units) */
} orc_ctx_t;

ORC_API orc_ctx_t *orc_ctx_create(int H, int W, int k, const float *K, const float *Kinv,
                                  const float *ref, const float *Rref, const float *tref,
                                  int S, const float *src, const float *Rs, const float *ts)
{
    if (S > ORC_MAX_SRC) return NULL;
    orc_ctx_t *c = (orc_ctx_t *)calloc(1, sizeof(orc_ctx_t));
    size_t n = (size_t)H * W;
    c->cam.H = H; c->cam.W = W; c->S = S; c->k = k;
    memcpy(c->cam.K, K, 36); memcpy(c->cam.Kinv, Kinv, 36);
    memcpy(c->cam.Rref, Rref, 36); memcpy(c->cam.tref, tref, 12);
    c->ref = ref;
    for (int s = 0; s < S; ++s) {
        c->src[s] = src + (size_t)s * n;
        memcpy(c->Rs[s], Rs + 9 * s, 36);
        memcpy(c->ts[s], ts + 3 * s, 12);
    }
    float *buf = (float *)malloc(sizeof(float) * n * 10);
    c->mean1 = buf; c->var1 = buf + n; c->sampled = buf + 2 * n; c->tmp = buf + 3 * n;
    c->bv = buf + 4 * n; c->bvv = buf + 5 * n; c->brv = buf + 6 * n; c->cost_s = buf + 7 * n;
    c->total = buf + 8 * n; c->count = buf + 9 * n;
    c->valid = (unsigned char *)malloc(n);
    orc_box_stats(ref, H, W, k, c->mean1, c->var1);
    return c;
}

ORC_API void orc_ctx_destroy(orc_ctx_t *c)
{
    if (!c) return;
    free(c->ref_code); free(c->ref_codef);              /* src codes share ref_code's block */
    free(c->mean1); free(c->valid); free(c);
}

/* ------------------------------------------------------------ fast mode -- */
/* M = K R_s R_ref^T K^-1 and b = K (t_s - R_s R_ref^T t_ref), all in double from the float32
 * operands, rounded to float32 once.  With them  [u z, v z, z]^T = d * M [x,y,1]^T + b  is the
 * chain back-project -> world -> source camera -> pixel of mvs_patchmatch.py:341-360 in one
 * affine map per source.  Sums run left to right; K^-1 by cofactors. */
static void fast_compose(const float *K, const float *Rr, const float *tr,
                         const float *Rs, const float *ts, float *M, float *b)
{
    double Kd[9], Ki[9], Rrel[9], trel[3], A[9];
    for (int i = 0; i < 9; ++i) Kd[i] = (double)K[i];
    const double det = Kd[0] * (Kd[4] * Kd[8] - Kd[5] * Kd[7]) - Kd[1] * (Kd[3] * Kd[8] - Kd[5] * Kd[6]) +
                       Kd[2] * (Kd[3] * Kd[7] - Kd[4] * Kd[6]);
    Ki[0] = (Kd[4] * Kd[8] - Kd[5] * Kd[7]) / det; Ki[1] = (Kd[2] * Kd[7] - Kd[1] * Kd[8]) / det;
    Ki[2] = (Kd[1] * Kd[5] - Kd[2] * Kd[4]) / det; Ki[3] = (Kd[5] * Kd[6] - Kd[3] * Kd[8]) / det;
    Ki[4] = (Kd[0] * Kd[8] - Kd[2] * Kd[6]) / det; Ki[5] = (Kd[2] * Kd[3] - Kd[0] * Kd[5]) / det;
    Ki[6] = (Kd[3] * Kd[7] - Kd[4] * Kd[6]) / det; Ki[7] = (Kd[1] * Kd[6] - Kd[0] * Kd[7]) / det;
    Ki[8] = (Kd[0] * Kd[4] - Kd[1] * Kd[3]) / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rrel[3 * i + j] = ((double)Rs[3 * i] * (double)Rr[3 * j] + (double)Rs[3 * i + 1] * (double)Rr[3 * j + 1]) +
                              (double)Rs[3 * i + 2] * (double)Rr[3 * j + 2];
    for (int i = 0; i < 3; ++i)
        trel[i] = (double)ts[i] - ((Rrel[3 * i] * (double)tr[0] + Rrel[3 * i + 1] * (double)tr[1]) +
                                   Rrel[3 * i + 2] * (double)tr[2]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = (Kd[3 * i] * Rrel[j] + Kd[3 * i + 1] * Rrel[3 + j]) + Kd[3 * i + 2] * Rrel[6 + j];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            M[3 * i + j] = (float)((A[3 * i] * Ki[j] + A[3 * i + 1] * Ki[3 + j]) + A[3 * i + 2] * Ki[6 + j]);
        b[i] = (float)((Kd[3 * i] * trel[0] + Kd[3 * i + 1] * trel[1]) + Kd[3 * i + 2] * trel[2]);
    }
}

/* Select the arithmetic mode of every later call on this context.  Fast mode needs 8-bit images
 * (every pixel exactly code/255): returns 0 on success, -1 if an image is not 8-bit exact. */
ORC_API int orc_ctx_set_mode(orc_ctx_t *c, int mode)
{
    if (!mode) { c->mode = 0; return 0; }
    const int H = c->cam.H, W = c->cam.W, k = c->k, h = k / 2;
    const size_t n = (size_t)H * W;
    if (!c->ref_code) {
        unsigned char *codes = (unsigned char *)malloc(n * (size_t)(c->S + 1));
        for (int v = 0; v <= c->S; ++v) {
            const float *img = v == 0 ? c->ref : c->src[v - 1];
            unsigned char *dst = codes + (size_t)v * n;
            for (size_t i = 0; i < n; ++i) {
                float q = rintf(img[i] * 255.0f);
                q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
                dst[i] = (unsigned char)q;
                if ((float)dst[i] / 255.0f != img[i]) { free(codes); return -1; }
            }
        }
        c->ref_code = codes;
        for (int s = 0; s < c->S; ++s) c->src_code[s] = codes + (size_t)(s + 1) * n;
        c->ref_codef = (float *)malloc(sizeof(float) * n * 3);
        c->m1f = c->ref_codef + n; c->v1f = c->ref_codef + 2 * n;
        for (size_t i = 0; i < n; ++i) c->ref_codef[i] = (float)codes[i];
        /* ref window sums: exact integers (<= k*k*255^2 < 2^24), any order */
        const float C1 = fast_c1(k), C2 = fast_c2(k);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int sr = 0, srr = 0;
                for (int i = -h; i <= h; ++i)
                    for (int j = -h; j <= h; ++j) {
                        int yy = y + i, xx = x + j;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            int r = codes[(size_t)yy * W + xx];
                            sr += r; srr += r * r;
                        }
                    }
                float m1 = (float)sr * C1;
                c->m1f[(size_t)y * W + x] = m1;
                c->v1f[(size_t)y * W + x] = fmaf(-m1, m1, (float)srr * C2);
            }
        for (int s = 0; s < c->S; ++s)
            fast_compose(c->cam.K, c->cam.Rref, c->cam.tref, c->Rs[s], c->ts[s], c->M[s], c->bvec[s]);
    }
    c->mode = 1;
    return 0;
}

static inline uint32_t f32_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }

/* Fast-mode projection + bilinear sample of source s at pixel (x,y), depth d; the result is in
 * code units (0..255).  bounds as in project_sample.  Validity: z > 0.1 and, with u' = u - lo,
 * 0 <= u' < W - 2 lo tested as ONE unsigned compare of the float's bit pattern (a negative or
 * NaN u' has a bit pattern above every non-negative bound); the same for v'. */
static inline float project_sample_fast(const orc_ctx_t *c, int s, int x, int y, float d,
                                        int half, int bounds, int *valid)
{
    const int H = c->cam.H, W = c->cam.W;
    const float *M = c->M[s], *b = c->bvec[s];
    const unsigned char *img = c->src_code[s];
    const float fx = (float)x, fy = (float)y;
    float q0 = fmaf(M[1], fy, fmaf(M[0], fx, M[2]));
    float q1 = fmaf(M[4], fy, fmaf(M[3], fx, M[5]));
    float q2 = fmaf(M[7], fy, fmaf(M[6], fx, M[8]));
    float p0 = fmaf(d, q0, b[0]), p1 = fmaf(d, q1, b[1]), p2 = fmaf(d, q2, b[2]);
    float zz = p2 + 1e-8f;
    float rz = 1.0f / zz;
    const int lo = bounds == 0 ? half : 0;
    float up = fmaf(p0, rz, -(float)lo), vp = fmaf(p1, rz, -(float)lo);
    int ok = p2 > 0.1f;
    if (bounds != 2)
        ok = ok && f32_bits(up) < f32_bits((float)(W - 2 * lo)) && f32_bits(vp) < f32_bits((float)(H - 2 * lo));
    *valid = ok;
    float x0 = floorf(up), y0 = floorf(vp);
    float wx = up - x0, wy = vp - y0;
    /* footprint origin clamped to true image coordinates [-2, W] x [-2, H]; NaN -> lower bound */
    const float xlo = -(float)(2 + lo), xhi = (float)(W - lo), yhi = (float)(H - lo);
    float xc = !(x0 >= xlo) ? xlo : (x0 > xhi ? xhi : x0);
    float yc = !(y0 >= xlo) ? xlo : (y0 > yhi ? yhi : y0);
    const int xi = (int)xc + lo, yi = (int)yc + lo;
    float t[4];
    for (int k = 0; k < 4; ++k) {
        int xx = xi + (k & 1), yy = yi + (k >> 1);
        t[k] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (float)img[(size_t)yy * W + xx] : 0.0f;
    }
    float top = fmaf(wx, t[1] - t[0], t[0]);
    float bot = fmaf(wx, t[3] - t[2], t[2]);
    return fmaf(wy, bot - top, top);
}

/* NCC of the context's reference image against a sampled image, in the context's mode */
static void ctx_ncc(orc_ctx_t *c, const float *sampled, int variant, float *out)
{
    const int H = c->cam.H, W = c->cam.W;
    if (c->mode)
        ncc_map_fast(c->ref_codef, c->m1f, c->v1f, sampled, H, W, c->k, variant, 0.0f, c->tmp, c->bv, c->bvv, c->brv, out);
    else
        ncc_map(c->ref, c->mean1, c->var1, sampled, H, W, c->k, variant, c->tmp, c->bv, c->bvv, c->brv, out);
}

/* sample source s at per-pixel depth map `depth` -> sampled image + validity */
static void sample_source(orc_ctx_t *c, int s, const float *depth, int bounds,
                          float *sampled, unsigned char *valid)
{
    const int H = c->cam.H, W = c->cam.W, half = c->k / 2;
    if (c->mode) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int ok;
                sampled[y * W + x] = project_sample_fast(c, s, x, y, depth[y * W + x], half, bounds, &ok);
                valid[y * W + x] = (unsigned char)ok;
            }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float Pw[3];
            int ok;
            backproject(&c->cam, x, y, depth[y * W + x], Pw);
            sampled[y * W + x] = project_sample(&c->cam, Pw, c->Rs[s], c->ts[s], c->src[s],
                                                half, bounds, &ok);
            valid[y * W + x] = (unsigned char)ok;
        }
}

/* test hook: sampled image + validity of one source (mvs_patchmatch.py:351-377) */
ORC_API void orc_sample(orc_ctx_t *c, int s, const float *depth, int bounds,
                        float *sampled_out, unsigned char *valid_out)
{
    sample_source(c, s, depth, bounds, sampled_out, valid_out);
}

/* _compute_patch_cost (mvs_patchmatch.py:323-390) */
ORC_API void orc_patch_cost(orc_ctx_t *c, const float *depth, float *cost_out)
{
    const int H = c->cam.H, W = c->cam.W;
    const size_t n = (size_t)H * W;
    (void)H; (void)W;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) { c->total[i] = 0.0f; c->count[i] = 0.0f; }
    for (int s = 0; s < c->S; ++s) {
        sample_source(c, s, depth, 0, c->sampled, c->valid);
        ctx_ncc(c, c->sampled, 0, c->cost_s);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i)
            if (c->valid[i]) { c->total[i] = c->total[i] + c->cost_s[i]; c->count[i] += 1.0f; }
    }
    const int fast = c->mode;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float cden = c->count[i] + 1e-8f;
        float avg = fast ? c->total[i] * (1.0f / cden) : qdiv(c->total[i], cden, 1.0f / cden);
        cost_out[i] = (c->count[i] >= 2.0f) ? avg : INFINITY;
    }
}

/* _compute_confidence (mvs_patchmatch.py:493-534) */
ORC_API void orc_confidence(orc_ctx_t *c, const float *depth, float *conf_out)
{
    const int H = c->cam.H, W = c->cam.W;
    const size_t n = (size_t)H * W;
    (void)H; (void)W;
    for (size_t i = 0; i < n; ++i) conf_out[i] = 0.0f;
    for (int s = 0; s < c->S; ++s) {
        sample_source(c, s, depth, 1, c->sampled, c->valid);
        ctx_ncc(c, c->sampled, 0, c->cost_s);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) {
            float ncc = 1.0f - c->cost_s[i];          /* :530 */
            if (c->valid[i] && ncc > 0.6f) conf_out[i] += 1.0f;
        }
    }
}

/* one pull step of _spatial_propagation (mvs_patchmatch.py:427-455): the
 * candidate at (y,x) is the state at (y+oy, x+ox); outside the image the
 * candidate depth is depth_min and the candidate normal is zero.           */
ORC_API void orc_propagate_step(orc_ctx_t *c, float *depth, float *normal, float *cost,
                                int oy, int ox, float depth_min)
{
    const int H = c->cam.H, W = c->cam.W;
    const size_t n = (size_t)H * W;
    float *cd = (float *)calloc(n * 5, sizeof(float));
    float *cn = cd + n, *cc = cd + 4 * n;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int yy = y + oy, xx = x + ox;
            size_t i = (size_t)y * W + x;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                size_t j = (size_t)yy * W + xx;
                cd[i] = depth[j];
                cn[3 * i] = normal[3 * j]; cn[3 * i + 1] = normal[3 * j + 1]; cn[3 * i + 2] = normal[3 * j + 2];
            } else {
                cd[i] = depth_min;
                cn[3 * i] = cn[3 * i + 1] = cn[3 * i + 2] = 0.0f;
            }
        }
    orc_patch_cost(c, cd, cc);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        if (cc[i] < cost[i]) {                                   /* :452-455 */
            depth[i] = cd[i]; cost[i] = cc[i];
            normal[3 * i] = cn[3 * i]; normal[3 * i + 1] = cn[3 * i + 1]; normal[3 * i + 2] = cn[3 * i + 2];
        }
    free(cd);
}

/* _spatial_propagation (mvs_patchmatch.py:415-457).  forward (even
 * iteration): offsets (-1,0),(0,-1) => F.pad pulls from (y+1,x) then (y,x+1);
 * backward: pulls from (y-1,x) then (y,x-1).                                */
ORC_API void orc_spatial_propagation(orc_ctx_t *c, float *depth, float *normal, float *cost,
                                     int forward, float depth_min)
{
    if (forward) {
        orc_propagate_step(c, depth, normal, cost, 1, 0, depth_min);
        orc_propagate_step(c, depth, normal, cost, 0, 1, depth_min);
    } else {
        orc_propagate_step(c, depth, normal, cost, -1, 0, depth_min);
        orc_propagate_step(c, depth, normal, cost, 0, -1, depth_min);
    }
}

/* one sample of _random_refinement (mvs_patchmatch.py:470-489) with the
 * noise tensors given explicitly (u: rand(H,W); nz: randn(H,W,3)).          */
ORC_API void orc_refine_step(orc_ctx_t *c, float *depth, float *normal, float *cost,
                             const float *u, const float *nz,
                             float depth_range, float normal_range,
                             float depth_min, float depth_max)
{
    const size_t n = (size_t)c->cam.H * c->cam.W;
    float *cd = (float *)calloc(n * 5, sizeof(float));
    float *cn = cd + n, *cc = cd + 4 * n;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float delta = (u[i] * 2.0f - 1.0f) * depth_range;        /* :471 */
        float d = depth[i] + delta;                              /* :472 */
        d = d < depth_min ? depth_min : d;
        d = d > depth_max ? depth_max : d;
        cd[i] = d;
        float a = normal[3 * i] + nz[3 * i] * normal_range;      /* :475-476 */
        float b = normal[3 * i + 1] + nz[3 * i + 1] * normal_range;
        float e = normal[3 * i + 2] + nz[3 * i + 2] * normal_range;
        normalize3(&a, &b, &e);
        cn[3 * i] = a; cn[3 * i + 1] = b; cn[3 * i + 2] = e;
    }
    orc_patch_cost(c, cd, cc);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        if (cc[i] < cost[i]) {                                   /* :486-489 */
            depth[i] = cd[i]; cost[i] = cc[i];
            normal[3 * i] = cn[3 * i]; normal[3 * i + 1] = cn[3 * i + 1]; normal[3 * i + 2] = cn[3 * i + 2];
        }
    free(cd);
}

/* init (mvs_patchmatch.py:268-284) from injected noise: u rand(H,W), n0/n1 randn(H,W) */
ORC_API void orc_init_state(int64_t n, const float *u, const float *n0, const float *n1,
                            float log_scale, float log_min,
                            float *depth, float *normal, float *cost)
{
    for (int64_t i = 0; i < n; ++i) {
        depth[i] = orc_expf(u[i] * log_scale + log_min);         /* :270-272 */
        float a = n0[i] * 0.3f, b = n1[i] * 0.3f, e = -1.0f;     /* :275-280 */
        normalize3(&a, &b, &e);
        normal[3 * i] = a; normal[3 * i + 1] = b; normal[3 * i + 2] = e;
        cost[i] = INFINITY;                                      /* :284 */
    }
}

/* _patchmatch_cuda (mvs_patchmatch.py:225-321) with the counter-hash RNG.
 * log_scale = (float)(ln depth_max - ln depth_min), log_min = (float)ln depth_min,
 * both formed in double by the caller as the reference does (:268-271).     */
ORC_API void orc_patchmatch_view(orc_ctx_t *c, int iters, int samples,
                                 float depth_min, float depth_max,
                                 float log_scale, float log_min,
                                 uint64_t seed, uint32_t view,
                                 float *depth, float *normal, float *conf)
{
    const int64_t n = (int64_t)c->cam.H * c->cam.W;
    float *cost = (float *)malloc(sizeof(float) * n * 5);
    float *u = cost + n, *nz = cost + 2 * n;
    float *n0 = (float *)malloc(sizeof(float) * n * 2), *n1 = n0 + n;
    orc_rng_fill(seed, view, 0, n, u, nz);
    for (int64_t i = 0; i < n; ++i) { n0[i] = nz[3 * i]; n1[i] = nz[3 * i + 1]; }
    orc_init_state(n, u, n0, n1, log_scale, log_min, depth, normal, cost);
    free(n0);
    for (int it = 0; it < iters; ++it) {
        orc_spatial_propagation(c, depth, normal, cost, (it % 2) == 0, depth_min);   /* :297 */
        /* :466-467  ranges are formed in double, then cast when they meet the tensor */
        float dr = (float)(((double)depth_max - (double)depth_min) * pow(0.5, it));
        float nr = (float)(0.5 * pow(0.5, it));
        for (int s = 0; s < samples; ++s) {
            orc_rng_fill(seed, view, (uint32_t)(1 + it * samples + s), n, u, nz);
            orc_refine_step(c, depth, normal, cost, u, nz, dr, nr, depth_min, depth_max);
        }
    }
    orc_confidence(c, depth, conf);                                                   /* :311 */
    free(cost);
}

/* _plane_sweep_torch (dense_stereo.py:222-316): D fronto-parallel planes,
 * vote (ncc > thresh) & (z > 0.1) per neighbour, torch.max(dim=0) keeps the
 * FIRST maximal plane index (:307).                                         */
ORC_API void orc_plane_sweep(orc_ctx_t *c, const float *depths, int D, float thresh,
                             float *depth_out, float *conf_out)
{
    const int H = c->cam.H, W = c->cam.W;
    const size_t n = (size_t)H * W;
    float *dmap = (float *)malloc(sizeof(float) * n * 2);
    float *votes = dmap + n;
    for (size_t i = 0; i < n; ++i) { conf_out[i] = -1.0f; depth_out[i] = 0.0f; }
    for (int d = 0; d < D; ++d) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) { dmap[i] = depths[d]; votes[i] = 0.0f; }
        for (int s = 0; s < c->S; ++s) {
            sample_source(c, s, dmap, 2, c->sampled, c->valid);
            if (c->mode && thresh > 0.0f) {
                /* fast mode: the vote through the squared comparison (ncc_map_fast, variant 2) */
                ncc_map_fast(c->ref_codef, c->m1f, c->v1f, c->sampled, H, W, c->k, 2, thresh * thresh,
                             c->tmp, c->bv, c->bvv, c->brv, c->cost_s);
#pragma omp parallel for schedule(static)
                for (size_t i = 0; i < n; ++i)
                    if (c->cost_s[i] != 0.0f && c->valid[i]) votes[i] += 1.0f;
                continue;
            }
            ctx_ncc(c, c->sampled, 1, c->cost_s);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; ++i)
                if (c->cost_s[i] > thresh && c->valid[i]) votes[i] += 1.0f;       /* :303-304 */
        }
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i)
            if (votes[i] > conf_out[i]) { conf_out[i] = votes[i]; depth_out[i] = depths[d]; }
    }
    free(dmap);
}

ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
