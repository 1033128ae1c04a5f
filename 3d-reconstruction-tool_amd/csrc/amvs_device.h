// amvs_device.h -- device-side arithmetic shared by the gfx950 kernels.
//
// All float32, compiled with -ffp-contract=off: the only fused operations are the
// explicit __builtin_fmaf calls, so results are reproducible bit for bit.
// Reference semantics: src/core/mvs_patchmatch.py (file:line cited per function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amvs_check.h"

#define AMVS_WAVE 64
#define AMVS_DEV __device__ __forceinline__

namespace amvs {

// Correctly rounded reciprocal and square root without the compiler's IEEE expansions (which
// cost ~17 and ~22 plain VALU issue slots on gfx950, tools/shift_rate.hip).  For every float
// whose biased exponent lies in [32, 222] (2^-95 <= |x| < 2^96)
//     v_rcp_f32 + one FMA correction        == 1.0f / x        (both signs, 3.2e9 inputs)
//     v_rsq_f32 + Goldschmidt + FMA residual == sqrtf(x)        (x > 0, 1.6e9 inputs; NaN for x < 0)
// bit for bit -- verified EXHAUSTIVELY on MI355X (tools/lean_math.hip, tests/test_hip_parity.py::
// test_lean_math_exhaustive).  Outside that range (zero, denormals, huge, negative for sqrt,
// NaN/inf) the IEEE expansion runs, behind a wave-uniform branch that is almost never taken.
AMVS_DEV float rcp_rn(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(r, e, r);
    const bool ok = (((__float_as_uint(x) >> 23) & 0xFFu) - 32u) <= 190u;
    if (__builtin_expect(!__all(ok), 0)) r = ok ? r : 1.0f / x;
    return r;
}

AMVS_DEV float sqrt_rn(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    // +-0 (all-zero sample windows are common near image borders) is returned as is; a negative
    // in-range argument already produced NaN through v_rsq_f32, as sqrtf does
    g = x == 0.0f ? x : g;
    const bool ok = ((((__float_as_uint(x) >> 23) & 0xFFu) - 32u) <= 190u) | (x == 0.0f);
    if (__builtin_expect(!__all(ok), 0)) g = ok ? g : __builtin_sqrtf(x);
    return g;
}

// 1 / (sqrt(x) + 1e-8) pieces of the NCC denominator (mvs_patchmatch.py:409-411) with ONE validity
// test: when x passes the square root's test (2^-95 <= x < 2^96, or zero) the denominator lies in
// [1e-8, 2^48], always inside the reciprocal's verified range, so the reciprocal needs no test of
// its own; otherwise both run as IEEE operations.
AMVS_DEV void ncc_denominator(float x, float &den, float &rden)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    g = x == 0.0f ? x : g;
    den = g + 1e-8f;
    float q = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, q, 1.0f);
    rden = __builtin_fmaf(q, e, q);
    const bool ok = ((((__float_as_uint(x) >> 23) & 0xFFu) - 32u) <= 190u) | (x == 0.0f);
    if (__builtin_expect(!__all(ok), 0)) {
        const float den_ieee = __builtin_sqrtf(x) + 1e-8f;
        den = ok ? den : den_ieee;
        rden = ok ? rden : 1.0f / den_ieee;
    }
}

// Optimistic forms for the hot loops: the lean result is always computed and the validity of the
// operand is ANDed into `ok` instead of branching; the caller tests `ok` once per stage and row
// (one wave-uniform branch instead of one per call) and, when some lane saw an operand outside the
// verified range, repeats the stage with LEAN = false (plain IEEE operations).  The stages are
// pure functions of their inputs, so the repeat is safe; it is practically never taken.
AMVS_DEV bool lean_exp_ok(float x) { return (((__float_as_uint(x) >> 23) & 0xFFu) - 32u) <= 190u; }

template <bool LEAN>
AMVS_DEV float rcp_t(float x, bool &ok)
{
    if constexpr (LEAN) {
        float r = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(r, e, r);
        ok &= lean_exp_ok(x);
        return r;
    } else {
        return 1.0f / x;
    }
}

template <bool LEAN>
AMVS_DEV float sqrt_t(float x, bool &ok)
{
    if constexpr (LEAN) {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y;
        float h = 0.5f * y;
        const float r = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, r, g);
        h = __builtin_fmaf(h, r, h);
        const float d = __builtin_fmaf(-g, g, x);
        g = __builtin_fmaf(d, h, g);
        g = x == 0.0f ? x : g;
        ok &= lean_exp_ok(x) | (x == 0.0f);
        return g;
    } else {
        return __builtin_sqrtf(x);
    }
}

// ---------------------------------------------------------------- RNG ------
// Counter-hash generator standing in for torch.rand / torch.randn
// (mvs_patchmatch.py:271,279,280,471,475).  A "draw" gives every pixel one
// uniform and three normals; streams are keyed by (seed, view, draw).
AMVS_DEV uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

struct StreamKey { uint32_t k1, k2; };

AMVS_DEV StreamKey stream_key(uint64_t seed, uint32_t view, uint32_t draw)
{
    uint32_t a = fmix32((uint32_t)seed ^ 0x9E3779B9u);
    a = fmix32(a + view);
    uint32_t b = fmix32((uint32_t)(seed >> 32) ^ 0x85EBCA6Bu);
    b = fmix32(b + draw);
    b = fmix32(b ^ a);
    return {a, b};
}

AMVS_DEV uint32_t pixel_hash(uint32_t idx, StreamKey k) { return fmix32(fmix32(idx ^ k.k1) + k.k2); }

AMVS_DEV float rng_uniform(uint32_t h0) { return (float)(h0 >> 8) * 0x1p-24f; }

// ln(t) for t in [0.5, 65536): exponent split + degree-9 polynomial in (m - 1).
AMVS_DEV float log_poly(float t)
{
    uint32_t u = __float_as_uint(t);
    int e = (int)((u >> 23) & 0xFF) - 127;
    float m = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f, 1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f, 1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f, 2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = f + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

// sin / cos on [0, pi/2]: Taylor polynomials in th^2 (errors < 6e-8).
AMVS_DEV void sincos_quadrant(float th, float &s, float &c)
{
    float t2 = th * th;
    float ps = 1.6059043837e-10f;
    ps = __builtin_fmaf(ps, t2, -2.5052108385e-8f);
    ps = __builtin_fmaf(ps, t2, 2.7557319224e-6f);
    ps = __builtin_fmaf(ps, t2, -1.9841269841e-4f);
    ps = __builtin_fmaf(ps, t2, 8.3333333333e-3f);
    ps = __builtin_fmaf(ps, t2, -1.6666666667e-1f);
    s = __builtin_fmaf(ps * t2, th, th);
    float pc = 2.0876756988e-9f;
    pc = __builtin_fmaf(pc, t2, -2.7557319224e-7f);
    pc = __builtin_fmaf(pc, t2, 2.4801587302e-5f);
    pc = __builtin_fmaf(pc, t2, -1.3888888889e-3f);
    pc = __builtin_fmaf(pc, t2, 4.1666666667e-2f);
    pc = __builtin_fmaf(pc, t2, -0.5f);
    c = __builtin_fmaf(pc, t2, 1.0f);
}

// Box-Muller pair from one word: 16-bit radius index, 16-bit angle.
AMVS_DEV void normal_pair(uint32_t w, float &n0, float &n1)
{
    uint32_t a = w >> 16, b = w & 0xFFFFu;
    float t = (float)a + 0.5f;
    float lnu = log_poly(t) + (-11.090354888959125f);
    float r = sqrt_rn(-2.0f * lnu);
    uint32_t q = b >> 14;
    float th = ((float)(b & 0x3FFFu) * 0x1p-14f) * 1.57079632679489662f;
    float s, c;
    sincos_quadrant(th, s, c);
    float cs = (q & 1u) ? s : c;
    float sn = (q & 1u) ? c : s;
    cs = (q == 1u || q == 2u) ? -cs : cs;
    sn = (q >= 2u) ? -sn : sn;
    n0 = r * cs; n1 = r * sn;
}

AMVS_DEV void rng_normals3(uint32_t h0, float &n0, float &n1, float &n2)
{
    float spare;
    normal_pair(fmix32(h0 + 0x9E3779B9u), n0, n1);
    normal_pair(fmix32(h0 + 0x3C6EF372u), n2, spare);
}

// exp for the log-uniform depth initialisation (mvs_patchmatch.py:270-272).
AMVS_DEV float exp_poly(float x)
{
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    int ni = (int)n;
    ni = ni > 127 ? 127 : ni;
    ni = ni < -126 ? -126 : ni;
    return y * __uint_as_float((uint32_t)(ni + 127) << 23);
}

// a / b given rb = RN(1/b): q = a*rb, one FMA residual, one FMA correction.  With a
// correctly rounded reciprocal this returns the correctly rounded quotient (Markstein);
// checked equal to IEEE division on 2.16e8 random operand pairs.  Used where
// several quotients share a denominator, so one division serves all of them.
AMVS_DEV float qdiv(float a, float b, float rb)
{
    float q = a * rb;
    float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, rb, q);
}

// F.normalize(v, dim=-1) = v / max(||v||, 1e-12)   (mvs_patchmatch.py:281,476)
AMVS_DEV void normalize3(float &x, float &y, float &z)
{
    float n = sqrt_rn(x * x + y * y + z * z);
    float d = n > 1e-12f ? n : 1e-12f;
    float rd = rcp_rn(d);
    x = qdiv(x, d, rd); y = qdiv(y, d, rd); z = qdiv(z, d, rd);
}

// ------------------------------------------------------------ geometry -----
struct Vec3 { float x, y, z; };

// rays = [x,y,1] @ K_inv.T ; X = rays * d ; Xw = (X - t_ref) @ R_ref
// (mvs_patchmatch.py:341-347).  3-term sums are fma(a2,b2,fma(a1,b1,a0*b0)), the order
// torch-CPU's matmul was measured to use.  Pointer types are templated so that loads from
// the constant address space stay scalar (s_load).
template <class KP, class RP, class TP>
AMVS_DEV Vec3 backproject(KP Kinv, RP Rref, TP tref, int x, int y, float d)
{
    float px = (float)x, py = (float)y;
    float q0 = __builtin_fmaf(1.0f, Kinv[2], __builtin_fmaf(py, Kinv[1], px * Kinv[0])) * d - tref[0];
    float q1 = __builtin_fmaf(1.0f, Kinv[5], __builtin_fmaf(py, Kinv[4], px * Kinv[3])) * d - tref[1];
    float q2 = __builtin_fmaf(1.0f, Kinv[8], __builtin_fmaf(py, Kinv[7], px * Kinv[6])) * d - tref[2];
    Vec3 w;
    w.x = __builtin_fmaf(q2, Rref[6], __builtin_fmaf(q1, Rref[3], q0 * Rref[0]));
    w.y = __builtin_fmaf(q2, Rref[7], __builtin_fmaf(q1, Rref[4], q0 * Rref[1]));
    w.z = __builtin_fmaf(q2, Rref[8], __builtin_fmaf(q1, Rref[5], q0 * Rref[2]));
    return w;
}

// Per-launch constants of the sampler.
struct SampleConsts {
    int H, W;
    float fw, fh;        // W-1, H-1
    float rfw, rfh;      // RN(1/(W-1)), RN(1/(H-1))
    float hw2, hh2;      // (W-1)/2, (H-1)/2
    float lo, hix, hiy;  // validity window of the projection
};

// wave-uniform float computed on the VALU -> SGPR (keeps loop-invariant constants out of the
// vector register budget; as an SGPR pair it also feeds packed operations directly)
AMVS_DEV float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

AMVS_DEV SampleConsts make_sample_consts(int H, int W, float lo, float hix, float hiy)
{
    SampleConsts c;
    c.H = H; c.W = W;
    const float fw = (float)(W - 1), fh = (float)(H - 1);
    c.fw = uniform_f(fw); c.fh = uniform_f(fh);
    c.rfw = uniform_f(1.0f / fw); c.rfh = uniform_f(1.0f / fh);
    c.hw2 = uniform_f(fw * 0.5f); c.hh2 = uniform_f(fh * 0.5f);
    c.lo = lo; c.hix = hix; c.hiy = hiy;
    return c;
}

// Sampling a source view is split into three phases so that a row can issue the gathers of
// ALL its sources back to back (one exposed memory latency per row instead of one per source):
//   sample_geom   project + bounds test + bilinear weights + addresses       (pure VALU)
//   sample_load   the gather(s)                                              (VMEM)
//   sample_finish decode + 4-tap fma chain                                   (LDS + VALU)
//
// Reference: mvs_patchmatch.py:351-377 (project, bounds test, grid_sample bilinear / zeros /
// align_corners=True), restated bit-exactly against ATen's CPU kernel.  The projection is
// valid when z > 0.1 and lo <= u < hix, lo <= v < hiy (patch bounds :362-363, image bounds
// :516-517, or -inf/+inf for the plane sweep, which only tests z: dense_stereo.py:280,303).
// `live` masks lanes whose pixel is outside the image: their gather reads element 0 and the
// sample is 0 (the zero padding of the box filter).
//
// U8 = false: the source is the float32 gray map; four dword gathers over two image rows.
// U8 = true : the source is the packed 8-bit row-pair map of an image whose every pixel equals
//             code/255 exactly (what cvtColor(...).astype(float32)/255 produces,
//             mvs_patchmatch.py:177): ushort (y,x) = code(y,x) | code(y+1,x) << 8, stored with a
//             TWO-texel border of zero codes on every side (row pitch W+4; code = 0 outside the
//             image).  One 2-byte-aligned dword gather fetches the whole 2x2 footprint with the
//             bytes at fixed positions, and the zero border IS grid_sample's zero padding: the
//             footprint origin is clamped to [-2, W] x [-2, H], where a tap outside the image reads
//             code 0 -> 0.0f, exactly what the reference's masked tap contributes -- no tap masks,
//             no byte-position selects.  Codes are decoded through the 256-entry table `lut` in
//             LDS (lut[c] = (float)c / 255.0f), so tap values are bit-identical to the float32 map's.
#define AMVS_PAIR_BORDER 2
// AMVS_CODE_BYTES = 1: the map holds ONE byte per texel (code(y,x), same zero border, row pitch W+4
// bytes) and a footprint is two 2-byte gathers (rows y0 and y0+1) instead of one 4-byte gather of
// a row pair: half the cache footprint per source image for one more load instruction.
#ifndef AMVS_CODE_BYTES
#define AMVS_CODE_BYTES 0
#endif
template <bool U8> struct TapGeom;
template <> struct TapGeom<true> {
    float nw, ne, sw, se;
    int off;            // ushort index of the 4-byte read from the first element of the padded map (>= 0)
};
template <> struct TapGeom<false> {
    float nw, ne, sw, se;
    int o00, o01, o10, o11;
    uint32_t sel;       // [27:24] tap masks
};
template <bool U8> struct TapRaw;
template <> struct TapRaw<true> { uint32_t w; };
template <> struct TapRaw<false> { float t00, t01, t10, t11; };

// NOBOUNDS (the plane sweep: lo = -inf, hix = hiy = +inf, dense_stereo.py:280,303 tests z only): the four bound
//          comparisons collapse to u < +inf, v < +inf (false exactly for NaN and +inf, as the four are).
// TRACK    (with LEAN): instead of testing every reciprocal's operand (`ok`), collect min / max |z + 1e-8| in
//          *zlo / *zhi; the caller tests the range once per row (the same condition, fewer instructions).
template <bool U8, bool LEAN, bool NOBOUNDS = false, bool TRACK = false, class KP, class RP, class TP>
AMVS_DEV TapGeom<U8> sample_geom(KP K, RP Rs, TP ts, const SampleConsts &c, Vec3 Pw, bool live, bool &valid,
                                 bool &ok, float *zlo = nullptr, float *zhi = nullptr)
{
    const int H = c.H, W = c.W;
    float p0 = __builtin_fmaf(Pw.z, Rs[2], __builtin_fmaf(Pw.y, Rs[1], Pw.x * Rs[0])) + ts[0];
    float p1 = __builtin_fmaf(Pw.z, Rs[5], __builtin_fmaf(Pw.y, Rs[4], Pw.x * Rs[3])) + ts[1];
    float z  = __builtin_fmaf(Pw.z, Rs[8], __builtin_fmaf(Pw.y, Rs[7], Pw.x * Rs[6])) + ts[2];
    float zz = z + 1e-8f;
    float rz;
    if constexpr (LEAN && TRACK) {
        rz = __builtin_amdgcn_rcpf(zz);
        rz = __builtin_fmaf(rz, __builtin_fmaf(-zz, rz, 1.0f), rz);
        const float az = __builtin_fabsf(zz);
        *zlo = __builtin_fminf(*zlo, az);
        *zhi = __builtin_fmaxf(*zhi, az);
    } else {
        rz = rcp_t<LEAN>(zz, ok);
    }
    float a = qdiv(p0, zz, rz), b = qdiv(p1, zz, rz);
    float u = __builtin_fmaf(b, K[1], a * K[0]) + K[2];
    float v = __builtin_fmaf(b, K[4], a * K[3]) + K[5];
    // non-short-circuit '&': '&&' makes hipcc emit a branch per source here
    if constexpr (NOBOUNDS) valid = (z > 0.1f) & (u < __builtin_inff()) & (v < __builtin_inff());
    else valid = (z > 0.1f) & (u >= c.lo) & (u < c.hix) & (v >= c.lo) & (v < c.hiy);
    float gx = qdiv(2.0f * u, c.fw, c.rfw) - 1.0f;
    float gy = qdiv(2.0f * v, c.fh, c.rfh) - 1.0f;
    float ux = (gx + 1.0f) * c.hw2;
    float uy = (gy + 1.0f) * c.hh2;
    float x0 = __builtin_floorf(ux), y0 = __builtin_floorf(uy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float wx1 = ux - x0, wx0 = x1 - ux, wy1 = uy - y0, wy0 = y1 - uy;
    TapGeom<U8> g;
    g.nw = wx0 * wy0; g.ne = wx1 * wy0; g.sw = wx0 * wy1; g.se = wx1 * wy1;
    if constexpr (U8) {
        // footprint origin clamped into the zero border (a NaN / infinite coordinate has NaN
        // weights, so which texels it reads does not matter)
        // (clamped as floats: one v_med3_f32 each, which also maps a NaN to the lower bound)
        const int cx = (int)__builtin_amdgcn_fmed3f(x0, -(float)AMVS_PAIR_BORDER, c.fw + 1.0f);
        const int cy = (int)__builtin_amdgcn_fmed3f(y0, -(float)AMVS_PAIR_BORDER, c.fh + 1.0f);
        // unsigned ushort index from the first element of the padded map (<= 2^29 + small)
        // (24-bit multiply: rows and pitch are far below 2^24; the full 32-bit one is a 64-bit mad)
        const unsigned idx = __umul24((unsigned)(cy + AMVS_PAIR_BORDER), (unsigned)(W + 2 * AMVS_PAIR_BORDER)) +
                             (unsigned)(cx + AMVS_PAIR_BORDER);
        g.off = AMVS_IDX(live ? (int)idx : 0, (H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - 1);   // (a footprint is 2 ushorts)
    } else {
        const int x0i = (int)x0, y0i = (int)y0;       // saturating conversion; NaN -> 0
        const bool x0ok = (x0 >= 0.0f) & (x0 <= c.fw), x1ok = (x1 >= 0.0f) & (x1 <= c.fw);
        const bool y0ok = (y0 >= 0.0f) & (y0 <= c.fh), y1ok = (y1 >= 0.0f) & (y1 <= c.fh);
        g.sel = ((x0ok & y0ok) ? 1u << 24 : 0u) | ((x1ok & y0ok) ? 1u << 25 : 0u) |
                ((x0ok & y1ok) ? 1u << 26 : 0u) | ((x1ok & y1ok) ? 1u << 27 : 0u);
        const int ix0 = min(max(x0i, 0), W - 1), ix1 = min(max((int)x1, 0), W - 1);
        const int iy0 = min(max(y0i, 0), H - 1), iy1 = min(max((int)y1, 0), H - 1);
        g.o00 = AMVS_IDX(live ? iy0 * W + ix0 : 0, H * W); g.o01 = AMVS_IDX(live ? iy0 * W + ix1 : 0, H * W);
        g.o10 = AMVS_IDX(live ? iy1 * W + ix0 : 0, H * W); g.o11 = AMVS_IDX(live ? iy1 * W + ix1 : 0, H * W);
    }
    return g;
}

// `img` is a device address from the job table: typed as a global pointer here, or the compiler
// would have to emit FLAT loads (which also count on the LDS/scalar wait counter)
typedef const __attribute__((address_space(1))) char *GlobalBytes;
typedef const __attribute__((address_space(1))) float *GlobalFloats;
typedef const __attribute__((address_space(1))) uint16_t *GlobalU16;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2_t *GlobalFloat2s;

// The four codes of a footprint whose origin is texel index `off` (non-negative) of the padded map
// starting at `img`: uniform 64-bit base + 32-bit byte offset, i.e. the scalar-base form of
// global_load.  Row-pair layout: one dword, bytes (y,x) (y+1,x) (y,x+1) (y+1,x+1); byte layout: two
// ushorts `pitch` bytes apart, combined to (y,x) (y,x+1) (y+1,x) (y+1,x+1).
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
AMVS_DEV uint32_t load_pair_word(unsigned long long img, int off, int pitch)
{
#if AMVS_CODE_BYTES
    us2_t w;
    unsigned short lo, hi;
    __builtin_memcpy(&lo, (GlobalBytes)img + (unsigned long long)(unsigned)off, 2);
    __builtin_memcpy(&hi, (GlobalBytes)img + (unsigned long long)(unsigned)(off + pitch), 2);
    w.x = lo; w.y = hi;
    return __builtin_bit_cast(uint32_t, w);
#else
    (void)pitch;
    uint32_t w;
    __builtin_memcpy(&w, (GlobalBytes)img + (unsigned long long)(2u * (unsigned)off), 4);
    return w;
#endif
}

template <bool U8>
AMVS_DEV TapRaw<U8> sample_load(unsigned long long img, const TapGeom<U8> &g, int pitch)
{
    TapRaw<U8> r;
    if constexpr (U8) {
        r.w = load_pair_word(img, g.off, pitch);
    } else {
        const GlobalFloats f = (GlobalFloats)img;
        r.t00 = f[g.o00]; r.t01 = f[g.o01]; r.t10 = f[g.o10]; r.t11 = f[g.o11];
    }
    return r;
}

template <bool U8>
AMVS_DEV float sample_finish(const TapRaw<U8> &r, const TapGeom<U8> &g, const float *lut, bool live)
{
    float t00, t01, t10, t11;
    if constexpr (U8) {
#if AMVS_CODE_BYTES
        t00 = lut[r.w & 0xFFu];
        t01 = lut[(r.w >> 8) & 0xFFu];
        t10 = lut[(r.w >> 16) & 0xFFu];
        t11 = lut[r.w >> 24];
#else
        t00 = lut[r.w & 0xFFu];
        t10 = lut[(r.w >> 8) & 0xFFu];
        t01 = lut[(r.w >> 16) & 0xFFu];
        t11 = lut[r.w >> 24];
#endif
    } else {
        t00 = (g.sel & (1u << 24)) ? r.t00 : 0.0f;
        t01 = (g.sel & (1u << 25)) ? r.t01 : 0.0f;
        t10 = (g.sel & (1u << 26)) ? r.t10 : 0.0f;
        t11 = (g.sel & (1u << 27)) ? r.t11 : 0.0f;
    }
    float v = __builtin_fmaf(t11, g.se, __builtin_fmaf(t10, g.sw, __builtin_fmaf(t01, g.ne, t00 * g.nw)));
    return live ? v : 0.0f;
}

// fill the 256-entry code -> gray table (one wave; entry c = (float)c / 255.0f, the
// float32 division numpy performs in `gray.astype(np.float32) / 255.0`)
AMVS_DEV void fill_gray_lut(float *lut, int lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) lut[lane * 4 + j] = (float)(lane * 4 + j) / 255.0f;
    __syncthreads();
}

// lane i <- lane i+1 (whole-wave shift; lane 63 receives 0).  hipcc folds this into the consuming VALU
// op as `v_add_f32_dpp ... wave_shl:1`.  (Keeping the shift a v_mov_b32_dpp of its own -- in isolation
// mov_dpp 1.84 ns + add 0.95 ns against 3.26 ns for the folded add, tools/valu_rate.hip -- was measured
// in the kernels in round 3: plane sweep 61.7 against 68.6 G px-hyp/s, pm_step 40.7 against 41.5: the
// folded form stays.)
AMVS_DEV float wave_shl1(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));
}

}  // namespace amvs
