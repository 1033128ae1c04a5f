// amvs_fast_common.h -- device code shared by the translation units of the FAST (tolerance) arithmetic:
// amvs_kernels_fast.hip (sweep step, sampling step, statistics) and amvs_sweep_fast.hip (plane sweep).
// See the header of amvs_kernels_fast.hip for what the fast arithmetic is.
#pragma once
#include "amvs_kernel_common.h"

namespace amvs {

// sources whose vertical ring lives in LDS (the others in shifting register rings)
#ifndef AMVS_FAST_RING_LDS
#define AMVS_FAST_RING_LDS 3
#endif
// sources sharing one opaque job-pointer copy (scheduling barrier): their geometry may interleave
#ifndef AMVS_FAST_RELOAD_STRIDE
#define AMVS_FAST_RELOAD_STRIDE 2
#endif
#ifndef AMVS_FAST_MIN_WAVES_BIAS
#define AMVS_FAST_MIN_WAVES_BIAS 0
#endif
// Resident workgroups per CU of the sweep step (StepArgs::wg_cap), enforced through unused dynamic
// LDS (160 KiB / (static + extra)).  Fewer resident waves touch fewer source rows at once: measured on
// MI355X (16 views 1080p, k=7, S=4, ms per launch, whole-schedule mean) 6 workgroups = 24 waves per CU
// 0.897, 5: 0.822, 4: 0.814, 3: 0.856 -- the launch is bound by the CU's L1 line rate for scattered
// gathers (2 cycles per distinct 128-byte line, tools/gather_rate.hip), not by latency, so the extra
// waves only add L2 misses.

// (21 x 21 and up: two sources in LDS, see ring_lds_sources in amvs_exact_common.h)
constexpr int fast_ring_lds(int K) { return K <= 19 ? AMVS_FAST_RING_LDS : 2; }
template <int K, int S> struct FRing {
    static constexpr int NL = fast_ring_lds(K) < S ? fast_ring_lds(K) : S;
    static constexpr int NR = S - NL > 0 ? S - NL : 1;
};


// the last K reference codes of a lane's column as packed bytes: the window occupies the TOP K
// bytes of NB dwords (oldest first)
template <int K> struct RefBytes {
    static constexpr int NB = (K + 3) / 4;
    static constexpr int FIRST = 4 * NB - K;       // byte index of the oldest window entry
};

template <int K>
AMVS_DEV void ref_bytes_push(uint32_t (&rb)[RefBytes<K>::NB], uint32_t code)
{
    constexpr int NB = RefBytes<K>::NB;
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) rb[i] = __builtin_amdgcn_alignbyte(rb[i + 1], rb[i], 1);
    rb[NB - 1] = __builtin_amdgcn_alignbyte(code, rb[NB - 1], 1);
}

template <int K>
AMVS_DEV float ref_bytes_get(const uint32_t (&rb)[RefBytes<K>::NB], int i)
{
    const int j = RefBytes<K>::FIRST + i;
    return (float)((rb[j >> 2] >> (8 * (j & 3))) & 0xFFu);          // v_cvt_f32_ubyteN
}

// wave-uniform constants of the fast sampler
struct FastConsts {
    float flo;                    // lower validity bound lo (patch half, or 0 for the confidence pass)
    uint32_t rxb, ryb;            // bit patterns of (float)(W - 2 lo), (float)(H - 2 lo)
    float cl_lo, cl_hix, cl_hiy;  // clamp of the footprint origin in (u - lo, v - lo) coordinates
    int pitch2;                   // bytes per row of the padded map
    int addc2;                    // byte offset of footprint origin (-(B+lo), -(B+lo)) ... see fast_geom
    int max_off;                  // byte offsets of a footprint's dword lie in [0, max_off) (index-checked build only)
};

AMVS_DEV FastConsts make_fast_consts(int H, int W, int lo)
{
    constexpr int B = AMVS_PAIR_BORDER;
    FastConsts c;
    c.flo = uniform_f((float)lo);
    c.rxb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint((float)(W - 2 * lo)));
    c.ryb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint((float)(H - 2 * lo)));
    c.cl_lo = uniform_f(-(float)(B + lo));
    c.cl_hix = uniform_f((float)(W - lo));
    c.cl_hiy = uniform_f((float)(H - lo));
    c.pitch2 = 2 * (W + 2 * B);
    c.addc2 = 2 * (lo + B) * (W + 2 * B + 1);
    c.max_off = 2 * (H + 2 * B) * (W + 2 * B) - 3;
    return c;
}

struct FastTap { float wx, wy; int off; };

// Projection of one pixel into one source (see the file header): weights, byte offset of the
// footprint's dword, validity.  LEAN: v_rcp_f32 + one FMA correction == 1.0f / zz wherever
// 2^-95 <= |zz| < 2^96 (amvs_device.h, verified exhaustively); the caller collects min / max |zz|
// over the sources and repeats the row with LEAN = false if a lane left that range.
// The column part of M [x, y, 1] -- fma(M0, x, M2), fma(M3, x, M5), fma(M6, x, M8) -- does not change
// along a lane's column: the row loops compute it once per strip and source (fast_column_terms)
// and pass it in; the operations and their order are those of the unhoisted form, so are the bits.
struct FastCol { float t0, t1, t2; };

template <class MP>
AMVS_DEV FastCol fast_column_terms(MP M, float fx)
{
    FastCol c;
    c.t0 = __builtin_fmaf(M[0], fx, M[2]);
    c.t1 = __builtin_fmaf(M[3], fx, M[5]);
    c.t2 = __builtin_fmaf(M[6], fx, M[8]);
    return c;
}

// TRACK: collect min / max |z| for the caller's range test (the plane sweep decides once per strip and
// plane instead, see plane_sweep_fast_kernel).
template <bool LEAN, bool BOUNDED, bool TRACK, class MP, class BP>
AMVS_DEV FastTap fast_geom(MP M, BP b, const FastConsts &fc, const FastCol &col, float fy, float d, bool &valid,
                           float &zlo, float &zhi)
{
    const float q0 = __builtin_fmaf(M[1], fy, col.t0);
    const float q1 = __builtin_fmaf(M[4], fy, col.t1);
    const float q2 = __builtin_fmaf(M[7], fy, col.t2);
    const float p0 = __builtin_fmaf(d, q0, b[0]);
    const float p1 = __builtin_fmaf(d, q1, b[1]);
    const float p2 = __builtin_fmaf(d, q2, b[2]);
    const float zz = p2 + 1e-8f;
    float rz;
    if constexpr (LEAN) {
        rz = __builtin_amdgcn_rcpf(zz);
        rz = __builtin_fmaf(rz, __builtin_fmaf(-zz, rz, 1.0f), rz);
        if constexpr (TRACK) {
            const float az = __builtin_fabsf(zz);
            zlo = __builtin_fminf(zlo, az);
            zhi = __builtin_fmaxf(zhi, az);
        }
    } else {
        rz = 1.0f / zz;
    }
    const float up = __builtin_fmaf(p0, rz, -fc.flo);
    const float vp = __builtin_fmaf(p1, rz, -fc.flo);
    valid = p2 > 0.1f;
    if constexpr (BOUNDED) {
        // non-short-circuit: '&&' makes hipcc emit a branch per source here
        const bool uin = __float_as_uint(up) < fc.rxb, vin = __float_as_uint(vp) < fc.ryb;
        valid = (bool)((int)valid & (int)uin & (int)vin);
    }
    const float x0 = __builtin_floorf(up), y0 = __builtin_floorf(vp);
    FastTap t;
    t.wx = up - x0;
    t.wy = vp - y0;
    // footprint origin clamped into the zero border (true coordinates [-2, W] x [-2, H]); v_med3_f32
    // maps a NaN to the lower bound
    const int xi = (int)__builtin_amdgcn_fmed3f(x0, fc.cl_lo, fc.cl_hix);
    const int yi = (int)__builtin_amdgcn_fmed3f(y0, fc.cl_lo, fc.cl_hiy);
    // byte offset from the first element of the padded map: ((yi+lo+B) * pitch + xi+lo+B) * 2 >= 0
    t.off = AMVS_IDX(__mul24(yi, fc.pitch2) + fc.addc2 + (xi << 1), fc.max_off);
    return t;
}

AMVS_DEV uint32_t fast_load(unsigned long long img, int off)
{
    uint32_t w;
    __builtin_memcpy(&w, (GlobalBytes)img + (unsigned long long)(unsigned)off, 4);
    return w;
}

// bytes of the dword: (y,x) (y+1,x) (y,x+1) (y+1,x+1); two horizontal lerps, one vertical
AMVS_DEV float fast_finish(uint32_t w, const FastTap &t, bool live)
{
    const float t00 = (float)(w & 0xFFu), t10 = (float)((w >> 8) & 0xFFu);
    const float t01 = (float)((w >> 16) & 0xFFu), t11 = (float)(w >> 24);
    const float top = __builtin_fmaf(t.wx, t01 - t00, t00);
    const float bot = __builtin_fmaf(t.wx, t11 - t10, t10);
    const float v = __builtin_fmaf(t.wy, bot - top, top);
    return live ? v : 0.0f;
}

// the column terms of all S sources for a lane's column (once per strip)
template <int S>
AMVS_DEV void fast_columns(JobCP job, float fx, FastCol (&cols)[S])
{
#pragma unroll
    for (int s = 0; s < S; ++s) cols[s] = fast_column_terms(job->fsrc[s].M, fx);
}

// PRIO (the sweep step): the wave raises its issue priority (s_setprio) from here until its gathers are
// requested.  Four waves share a SIMD; the arbiter then lets a wave that is forming its sample addresses
// go ahead of waves that are in their window sums, so the gathers of a row leave as early as possible and
// the memory pipe stays fed while the others' VALU work proceeds.  Measured on MI355X (config 3, same run,
// G px-hyp/s): 44.5-44.8 against 43.1-43.5 (+3.1 %); priority level 1, 2 or 3 and raising it already at
// the top of the row (before the state loads) make no difference; raising it for the window sums instead
// +1.6 %; the plane sweep (VALU-bound, coherent gathers) does not move (69.0 against 69.4 / 68.9).
template <int S, bool LEAN, bool BOUNDED, bool TRACK = LEAN, bool PRIO = false>
AMVS_DEV unsigned fast_sample_sources(JobCP job, const FastConsts &fc, const FastCol (&cols)[S], float fy, float d,
                                      bool live, float (&v)[S], bool &ok)
{
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    unsigned okbits = 0u;
    FastTap tg[S];
    uint32_t raw[S];
    float zlo = 1.0f, zhi = 1.0f;
    JobCP jr = job;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s % AMVS_FAST_RELOAD_STRIDE == 0) jr = reload(jr);
        float M[9], b[3];
        M[1] = jr->fsrc[s].M[1]; M[4] = jr->fsrc[s].M[4]; M[7] = jr->fsrc[s].M[7];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = jr->fsrc[s].b[i];
        const unsigned long long img = jr->fsrc[s].pairs;
        bool valid;
        tg[s] = fast_geom<LEAN, BOUNDED, TRACK>(M, b, fc, cols[s], fy, d, valid, zlo, zhi);
        okbits |= valid ? (1u << s) : 0u;
        raw[s] = fast_load(img, tg[s].off);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LEAN && TRACK) ok = (zlo >= 0x1p-95f) & (zhi < 0x1p96f);
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = fast_finish(raw[s], tg[s], live);
    return okbits;
}

// optimistic lean reciprocals first, IEEE repeat if some lane's z left the verified range
template <int S, bool BOUNDED, bool PRIO = false>
AMVS_DEV unsigned fast_sample_sources_checked(JobCP job, const FastConsts &fc, const FastCol (&cols)[S], float fy,
                                              float d, bool live, float (&v)[S])
{
    bool ok = true;
    unsigned okbits = fast_sample_sources<S, true, BOUNDED, true, PRIO>(job, fc, cols, fy, d, live, v, ok);
    if (__builtin_expect(!__all(ok), 0))
        okbits = fast_sample_sources<S, false, BOUNDED, false, PRIO>(reload(job), fc, cols, fy, d, live, v, ok);
    return okbits;
}

template <int K, int S>
AMVS_DEV void fring_push(float *lring, int lane, int wslot, float (&ring_v)[FRing<K, S>::NR][K], const float (&v)[S])
{
    constexpr int NL = FRing<K, S>::NL;
#pragma unroll
    for (int s = 0; s < NL; ++s) lring[(s * K + wslot) * AMVS_WAVE + lane] = v[s];
#pragma unroll
    for (int s = NL; s < S; ++s) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) ring_v[s - NL][i] = ring_v[s - NL][i + 1];
        ring_v[s - NL][K - 1] = v[s];
    }
}

// k x k window sums of v, v*v and r*v (code units) for S sources: column sums top -> bottom (plain
// sum for v, FMA chains for v*v and r*v), row sums right -> left as K-1 DPP wave shifts -- the
// order of the exact kernels (and of the tests' CPU checker).
// REV: the rings were filled walking UP the image (paired-band schedule, bottom-up wave): ring entry i is
// then row (K-1-i) of the window, and the column sums take them newest first -- the same top -> bottom
// order of the same values.
template <int K, int S, bool REV = false>
AMVS_DEV void window_sums_fast(const float *lring, int oldest, const float (&rr_in)[K],
                               const float (&ring_v)[FRing<K, S>::NR][K], int lane,
                               float (&bv)[S], float (&bvv)[S], float (&brv)[S])
{
    constexpr int NL = FRing<K, S>::NL;
    int slot[K];
    float rr[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = REV ? K - 1 - i : i;              // window row i (top -> bottom) = ring age j
        slot[i] = oldest + j >= K ? oldest + j - K : oldest + j;
        rr[i] = rr_in[j];
    }
    float cs[3 * S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float vv[K];
#pragma unroll
        for (int i = 0; i < K; ++i)
            vv[i] = s < NL ? lring[(s * K + slot[i]) * AMVS_WAVE + lane] : ring_v[s < NL ? 0 : s - NL][REV ? K - 1 - i : i];
        float cv = vv[0];
        float cvv = vv[0] * vv[0];
        float crv = rr[0] * vv[0];
#pragma unroll
        for (int i = 1; i < K; ++i) {
            cv = cv + vv[i];
            cvv = __builtin_fmaf(vv[i], vv[i], cvv);
            crv = __builtin_fmaf(rr[i], vv[i], crv);
        }
        cs[3 * s] = cv; cs[3 * s + 1] = cvv; cs[3 * s + 2] = crv;
    }
    float acc[3 * S];
#pragma unroll
    for (int i = 0; i < 3 * S; ++i) acc[i] = cs[i];
#pragma unroll
    for (int j = 1; j < K; ++j)
#pragma unroll
        for (int i = 0; i < 3 * S; ++i) acc[i] = wave_shl1(acc[i]) + cs[i];
#pragma unroll
    for (int s = 0; s < S; ++s) { bv[s] = acc[3 * s]; bvv[s] = acc[3 * s + 1]; brv[s] = acc[3 * s + 2]; }
}

}  // namespace amvs
