// amvs_kernels_fast.hip -- the sweep kernels in the FAST (tolerance) arithmetic, AMVS_MODE_FAST.
//
// Same algorithm, same execution shape and same RNG / candidate / select logic as
// amvs_kernels.hip (reference: src/core/mvs_patchmatch.py:323-534, src/core/dense_stereo.py:
// 262-310); what changes is the arithmetic of one cost evaluation, which no longer reproduces
// ATen's float32 operation sequence bit for bit but stays within the tolerances stated in
// DESIGN.md (and is itself restated operation for operation by the tests' CPU checker, so these
// kernels are still verified BIT-EXACTLY, against that restatement's fast mode):
//   * projection precomposed per (reference, source) pair on the host (FastSrc, amvs_kernels.h):
//     [u z, v z, z] = d * (M [x,y,1]) + b  -- 6 + 3 FMAs instead of two 3x3 rotations, two
//     translations and the intrinsics; one reciprocal of z; no Markstein quotient refinement; no
//     normalise / un-normalise round trip around grid_sample (mvs_patchmatch.py:367-377);
//   * validity u in [lo, W-lo) as ONE unsigned compare of the bit pattern of u - lo;
//   * the four codes of a footprint are converted with v_cvt_f32_ubyte{0..3} (no LDS table) and
//     interpolated as two horizontal lerps + one vertical lerp; all window sums run in code units
//     (0..255), the 1/255 and 1/k^2 factors are folded into two constants of the NCC epilogue;
//   * the reference image's window sums are exact integers, precomputed once per view and patch
//     size as (mean1, var1) maps (launch_fast_stats): no ref sums, and no ref ring in LDS -- the
//     last k reference codes of a column travel as packed bytes in 2-3 VGPRs;
//   * NCC epilogue: cov * RN(1/den) (no quotient refinement).
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

template <int S> struct FRing {
    static constexpr int NL = AMVS_FAST_RING_LDS < S ? AMVS_FAST_RING_LDS : S;
    static constexpr int NR = S - NL > 0 ? S - NL : 1;
};


template <int K, int S> struct StepLds {
    static constexpr unsigned PER_WAVE = (FRing<S>::NL > 0 ? FRing<S>::NL : 1) * K * AMVS_WAVE * 4u + 2u * AMVS_WAVE * 8u;
    static constexpr unsigned STATIC = AMVS_WG_WAVES * PER_WAVE;
    static constexpr unsigned XBUF = (K / 2) * S * AMVS_WAVE * 4u;                     // paired bands: the exchange rows of a wave
    static unsigned extra(int wg_cap, bool pair = false)
    {
        // wg_cap counts workgroups of AMVS_WG_WAVES waves; a paired workgroup of PAIR_WAVES waves takes
        // the share of PAIR_WAVES / AMVS_WG_WAVES of them
        const unsigned cap = (unsigned)(wg_cap > 0 ? wg_cap : AMVS_DEFAULT_WGS_PER_CU);
        const unsigned share = pair ? 160u * 1024u * PAIR_WAVES / (cap * AMVS_WG_WAVES) : 160u * 1024u / cap;
        const unsigned st = pair ? PAIR_WAVES * (PER_WAVE + XBUF) : STATIC;
        return st < share ? share - st : 0u;
    }
};

// the paired-band schedule is compiled where its exchange rows fit beside the rings at 4 workgroups per CU
constexpr bool fast_pair_supported(int K, int S)
{
    return K <= 7 && S <= 4;
}

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
    t.off = __mul24(yi, fc.pitch2) + fc.addc2 + (xi << 1);
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
AMVS_DEV void fring_push(float *lring, int lane, int wslot, float (&ring_v)[FRing<S>::NR][K], const float (&v)[S])
{
    constexpr int NL = FRing<S>::NL;
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
                               const float (&ring_v)[FRing<S>::NR][K], int lane,
                               float (&bv)[S], float (&bvv)[S], float (&brv)[S])
{
    constexpr int NL = FRing<S>::NL;
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

constexpr int fast_min_waves(int K, int S)
{
    return ((S + 1) * K <= 40 ? 4 : ((S + 1) * K <= 60 ? 3 : 2)) + AMVS_FAST_MIN_WAVES_BIAS;
}

// Depth hypothesis a pixel is sampled at in this step: the (offset) pixel's current depth for
// propagation / evaluation / confidence, a clamped random perturbation of it for refinement
// (mvs_patchmatch.py:430-436, :468-473).  `d_raw` is d_in at the pixel (+ offset) when `inb`.
AMVS_DEV float candidate_depth(const StepArgs &a, int mode, bool inb, float d_raw, uint32_t h0)
{
    const float dc = inb ? depth_untag(d_raw, a.depth_mask) : a.depth_min;
    const float delta = (rng_uniform(h0) * 2.0f - 1.0f) * a.depth_range;
    float d = dc + delta;
    d = d < a.depth_min ? a.depth_min : d;
    d = d > a.depth_max ? a.depth_max : d;
    return mode == MODE_REFINE ? d : dc;
}

// Sample maps of the split schedule (StepArgs::samples): [slot][source][H*W] floats, the sampled
// value in code units (>= 0 or NaN) with the validity of the projection in the sign bit (set = invalid).
AMVS_DEV uint32_t sample_encode(float v, bool ok)
{
    return (__float_as_uint(v) & 0x7FFFFFFFu) | (ok ? 0u : 0x80000000u);
}

// ------------------------------------------------------------------ sampling step ---
// First half of a split sweep step: every pixel of the launch's views once, NO strip halo --
// hypothesis, projection into the S sources, bilinear sample -> the sample maps.  The launch is
// bound by the CU's L1 line rate (2 cycles per gathered 128-byte line); the window / NCC / select
// half (pm_step_fast_kernel<..., PRE = true>) streams the sample maps and runs concurrently with
// the sampling half of another view group (amvs_capi.hip, run_split_schedule).
template <int S, int MODE_T>
__global__ __launch_bounds__(AMVS_WAVE * AMVS_WG_WAVES) void pm_sample_fast_kernel(const StepArgs a)
{
    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    const int per_job = a.s_tiles_x * a.s_tiles_y;
    const int t = xcd_remap(blockIdx.x, gridDim.x) * AMVS_WG_WAVES + wv;
    if (t >= a.n_jobs * per_job) return;                       // last workgroup only
    const int job_id = t / per_job;
    const int rem = t - job_id * per_job;
    const int ty = rem / a.s_tiles_x, tx = rem - ty * a.s_tiles_x;

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W, mode = MODE_T >= 0 ? MODE_T : a.mode;
    const long long HW = (long long)H * W;
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    uint32_t *__restrict__ out = (uint32_t *)a.samples + job->slot * HW * S;
    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);
    const FastConsts fc = make_fast_consts(H, W, mode == MODE_CONF ? 0 : a.half);

    const int xr = tx * AMVS_WAVE + lane;
    const int y0 = ty * a.s_TH;
    const float fx = (float)xr;
    FastCol cols[S];
    fast_columns<S>(job, fx, cols);
    const bool col_in = xr < W;
    const int rows = min(a.s_TH, H - y0);
    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    for (int r = 0; r < rows; ++r) {
#if AMVS_WG_WAVES > 1 && AMVS_WG_SYNC_ROWS > 0
        if (r % AMVS_WG_SYNC_ROWS == 0) __builtin_amdgcn_s_barrier();
#endif
        const int yr = y0 + r;
        const bool live = col_in;
        const bool inb = live & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
        const int pix = yr * W + xr;
        const float d_raw = d_in[inb ? pix + noff : 0];
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        const float dc = candidate_depth(a, mode, inb, d_raw, h0);
        float v[S];
        const unsigned okbits = fast_sample_sources_checked<S, true>(job, fc, cols, (float)yr, dc, live, v);
        if (live) {
#pragma unroll
            for (int s = 0; s < S; ++s) out[s * HW + pix] = sample_encode(v[s], (okbits >> s) & 1u);
        }
    }
}

// ------------------------------------------------------------------ sweep step ---
// PRE: the samples come from the sample maps written by pm_sample_fast_kernel (split schedule)
// instead of being gathered here; everything after the sampling stage is the same code.
// PAIR (StepArgs::paired, AMVS_SCHEDULE_PAIRED): a workgroup is 2 strip columns x 2 vertically adjacent
// bands.  The waves of the upper band walk DOWN, those of the lower band walk UP, so that both reach the
// common boundary at the same time; there they exchange the samples of their last K/2 rows through LDS
// and finish their last K/2 output rows from the partner's samples instead of sampling a halo of their own:
// K/2 halo rows per strip instead of K - 1 (the samples, and every sum over them in the same top -> bottom
// order, are those of the classic strips: bit-identical results).
template <int K, int S, int MODE_T, bool PRE = false, bool PAIR = false>
__global__ __launch_bounds__(AMVS_WAVE * (PAIR ? PAIR_WAVES : AMVS_WG_WAVES), fast_min_waves(K, S)) void pm_step_fast_kernel(const StepArgs a)
{
    constexpr int WGW = PAIR ? PAIR_WAVES : AMVS_WG_WAVES;      // waves of this workgroup
    static_assert(!(PRE && PAIR), "the paired bands sample for themselves");
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));      // 1 / (k^2 * 255)
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));    // 1 / (k^2 * 255^2)
    constexpr int NL = FRing<S>::NL;
    __shared__ float lring_all[WGW * (NL > 0 ? NL : 1) * K * AMVS_WAVE];
    constexpr int NQ = 2 * AMVS_WAVE;
    __shared__ uint2 nq_all[WGW * NQ];
    __shared__ float xbuf_all[PAIR ? WGW * HALF * S * AMVS_WAVE : 1];             // [wave][row][source][lane]

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    float *lring = lring_all + wv * ((NL > 0 ? NL : 1) * K * AMVS_WAVE);
    uint2 *nq = nq_all + wv * NQ;
    int q_head = 0, q_tail = 0;
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    int job_id, ty, tx;
    bool paired = false;                       // this wave has a partner band to exchange with
    int up = 0;                                // 1: the wave walks up the image (lower band of a pair)
    if constexpr (PAIR) {
        constexpr int PC = AMVS_PAIR_COLS;                      // strip columns of a workgroup
        const int col_pairs = (a.tiles_x + PC - 1) / PC, pair_rows = (a.tiles_y + 1) / 2;
        const int wg = xcd_remap(blockIdx.x, gridDim.x);
        job_id = wg / (col_pairs * pair_rows);
        const int rem = wg - job_id * (col_pairs * pair_rows);
        const int py = rem / col_pairs, px = rem - py * col_pairs;
        tx = PC * px + (wv % PC);
        up = wv / PC;
        ty = 2 * py + up;
        paired = 2 * py + 1 < a.tiles_y;
        if (job_id >= a.n_jobs || tx >= a.tiles_x || ty >= a.tiles_y) return;   // (the partner of an exiting wave exits too,
                                                                                //  or runs unpaired: `paired` is false)
    } else {
        const int t = xcd_remap(blockIdx.x, gridDim.x) * AMVS_WG_WAVES + wv;
        if (t >= a.n_jobs * tiles_per_job) return;             // last workgroup only
        strip_of(a, t, job_id, ty, tx);
    }

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W, mode = MODE_T >= 0 ? MODE_T : a.mode;
    const long long HW = (long long)H * W;
    constexpr int PADW = 2 * AMVS_PAIR_BORDER;
    // (global address space: a generic pointer would make these FLAT loads, which force vmcnt(0) and
    // lgkmcnt(0) waits)
    const GlobalU16 ref_pairs = (GlobalU16)job->ref_pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    float *__restrict__ d_out = a.d_out + job->slot * HW;
    float *cost_io = a.cost + job->slot * HW;
    float *nbuf0 = a.nbuf[0] + job->slot * HW * 3, *nbuf1 = a.nbuf[1] + job->slot * HW * 3;
    float *__restrict__ aux = a.aux + job->slot * HW;
    const uint32_t *__restrict__ smp = PRE ? (const uint32_t *)a.samples + job->slot * HW * S : nullptr;

    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);
    // validity window of the projection: patch bounds (mvs_patchmatch.py:362-363) or image bounds
    // for the confidence pass (:516-517)
    const FastConsts fc = make_fast_consts(H, W, mode == MODE_CONF ? 0 : HALF);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const float fx = (float)xr;
    FastCol cols[S];
    fast_columns<S>(job, fx, cols);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int th_w = min(a.TH, H - y0);                        // output rows of this strip
    // classic: all strips of a workgroup lie in one band and walk th_w + K - 1 rows.  PAIR: every wave of
    // the workgroup runs the same a.TH + K - 1 steps (common barriers); a wave whose band is shorter idles
    // first, so that the partners meet at their boundary in the same step
    const int rows = PAIR ? a.TH + 2 * HALF : th_w + 2 * HALF;
    const int idle_first = (PAIR && up) ? a.TH - th_w : 0;
    const int n_loc = th_w + 2 * HALF;                         // steps this wave works
    const int n_own = paired ? th_w + HALF : n_loc;            // ... of which it samples itself
    const int y_start = up ? y0 + th_w + HALF - 1 : y0 - HALF, dy = up ? -1 : 1;
    float *xmine = PAIR ? xbuf_all + wv * (HALF * S * AMVS_WAVE) : nullptr;
    const float *xpartner = PAIR ? xbuf_all + (wv ^ AMVS_PAIR_COLS) * (HALF * S * AMVS_WAVE) : nullptr;

    uint32_t rb[RefBytes<K>::NB];
    float ring_v[FRing<S>::NR][K];
    typename Hist<K, S>::T hist_ok = 0;
    uint32_t hist_h0[HALF + 1];
#pragma unroll
    for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int s = 0; s < FRing<S>::NR; ++s) ring_v[s][i] = 0.0f;
#pragma unroll
    for (int i = 0; i <= HALF; ++i) hist_h0[i] = 0u;
    int wslot = 0;

    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    for (int r = 0; r < rows; ++r) {
#if AMVS_WG_WAVES > 1 && AMVS_WG_SYNC_ROWS > 0
        if (r % AMVS_WG_SYNC_ROWS == 0) __builtin_amdgcn_s_barrier();
#endif
        if constexpr (PAIR) {
            // the partners have written the samples of their last K/2 own rows (steps TH .. TH + K/2 - 1)
            if (r == a.TH + HALF) __syncthreads();
        }
        const int loc = r - idle_first;                        // this wave's step
        if (PAIR && (loc < 0 || loc >= n_loc)) continue;        // (wave-uniform)
        const int yr = PAIR ? y_start + dy * loc : y0 - HALF + r;
        const bool own = !PAIR || loc < n_own;                  // sampled here, not taken from the partner
        const bool live = col_in & ((unsigned)yr < (unsigned)H);
        const bool inb = live & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
        const int pix = yr * W + xr;
        const uint32_t rc_raw = ref_pairs[live ? pix + PADW * yr : 0];
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        float v[S];
        unsigned okbits = 0u;
        if (PAIR && !own) {
            // a row of the partner band: its samples, taken at its own candidates, from LDS (the partner
            // wrote them walking towards the boundary: the row next to it last)
            const float *xp = xpartner + (HALF - 1 - (loc - n_own)) * (S * AMVS_WAVE);
#pragma unroll
            for (int s = 0; s < S; ++s) v[s] = xp[s * AMVS_WAVE + lane];
        } else if constexpr (PRE) {
            const uint32_t *__restrict__ sp = smp + (live ? pix : 0);
            uint32_t w[S];
#pragma unroll
            for (int s = 0; s < S; ++s) w[s] = sp[s * HW];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                v[s] = live ? __uint_as_float(w[s] & 0x7FFFFFFFu) : 0.0f;
                okbits |= (w[s] >> 31) ? 0u : (1u << s);
            }
        } else {
            // ---- candidate depth of this (possibly halo) pixel: as in the exact kernel ----
            const float d_raw = d_in[inb ? pix + noff : 0];
            const float dc = candidate_depth(a, mode, inb, d_raw, h0);
            okbits = fast_sample_sources_checked<S, true, true>(job, fc, cols, (float)yr, dc, live, v);
            if constexpr (PAIR) {
                if (paired && loc >= n_own - HALF) {           // the last K/2 own rows: for the partner
                    float *xm = xmine + (loc - (n_own - HALF)) * (S * AMVS_WAVE);
#pragma unroll
                    for (int s = 0; s < S; ++s) xm[s * AMVS_WAVE + lane] = v[s];
                }
            }
        }
        const uint32_t rcode = live ? (rc_raw & 0xFFu) : 0u;

        // ---- push into the vertical rings ----
        ref_bytes_push<K>(rb, rcode);
        fring_push<K, S>(lring, lane, wslot, ring_v, v);
        wslot = wslot + 1 == K ? 0 : wslot + 1;
        hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
#pragma unroll
        for (int i = 0; i < HALF; ++i) hist_h0[i] = hist_h0[i + 1];
        hist_h0[HALF] = h0;

        if ((PAIR ? loc : r) < 2 * HALF) continue;

        // ---- window sums, NCC, aggregate for centre row yc and centre column xc ----
        const int yc = PAIR ? yr - dy * HALF : yr - HALF;
        const int xc = xr + HALF;
        const bool outl = (lane < OUTW) & (xc < W);
        const int pc = outl ? yc * W + xc : 0;
        const float oldd_tagged = d_in[pc], oldc = cost_io[pc];
        const float oldd = depth_untag(oldd_tagged, a.depth_mask);
        const unsigned buf_c = depth_buffer(oldd_tagged);     // where this pixel's current normal lives
        // propagation: the neighbour the candidate was pulled from (requested with the other state
        // loads: behind the selection it would expose a memory round trip in every row).  Lanes
        // without an output pixel have pc = 0: they must not form pc + noff, which lies BEFORE the map
        // for the negative offsets of odd iterations.
        const bool inb_c = ((unsigned)(yc + oy) < (unsigned)H) & ((unsigned)(xc + ox) < (unsigned)W);
        const int pn = (outl & inb_c) ? pc + noff : 0;
        const float nb_tagged = mode == MODE_PROP ? d_in[pn] : 0.0f;
        const f32x2_t mv1 = ref_stats[pc];
        const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);
        const uint32_t h0c = (uint32_t)__shfl_down((int)hist_h0[0], HALF);

        float rr[K];
#pragma unroll
        for (int i = 0; i < K; ++i) rr[i] = ref_bytes_get<K>(rb, i);
        float bvs[S], bvvs[S], brvs[S];
        if (PAIR && up) window_sums_fast<K, S, true>(lring, wslot, rr, ring_v, lane, bvs, bvvs, brvs);
        else window_sums_fast<K, S>(lring, wslot, rr, ring_v, lane, bvs, bvvs, brvs);
        const float m1 = mv1.x, v1 = mv1.y;

        float total = 0.0f, cnt = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // _ncc_cost (mvs_patchmatch.py:403-411), sums in code units
            const float mean2 = bvs[s] * C1;
            const float var2 = __builtin_fmaf(-mean2, mean2, bvvs[s] * C2);
            const float cov = __builtin_fmaf(-m1, mean2, brvs[s] * C2);
            float den, rden;
            ncc_denominator(v1 * var2, den, rden);
            const float cost = 1.0f - cov * rden;
            const bool oks = (okc >> s) & 1u;
            const float ncc2 = 1.0f - cost;                         // :530
            const bool hit = mode == MODE_CONF ? (oks & (ncc2 > 0.6f)) : oks;
            total = (hit & (mode != MODE_CONF)) ? total + cost : total;
            cnt = hit ? cnt + 1.0f : cnt;
        }
        const bool act = outl;

        if (mode == MODE_CONF) {
            if (act) aux[pc] = cnt;
            continue;
        }

        // average over valid sources, +inf when fewer than two (mvs_patchmatch.py:387-388)
        const float cden = cnt + 1e-8f;
        bool cden_ok = true;
        const float avg = total * rcp_t<true>(cden, cden_ok);
        const float newc = cnt >= 2.0f ? avg : __builtin_inff();
        if (mode == MODE_EVAL) {
            if (act) aux[pc] = newc;
            continue;
        }

        // ---- select (mvs_patchmatch.py:452-455 / :486-489): as in the exact kernel ----
        const bool better = act & (newc < oldc);
        if (better) cost_io[pc] = newc;
        if (mode == MODE_PROP) {
            // out-of-image neighbour: depth_min and a zero normal (F.pad, :431-442).  Only the winners'
            // normals move (StepArgs::nbuf), queued and moved 64 at a time like the refinement winners'.
            const float nb_d = depth_untag(nb_tagged, a.depth_mask);
            if (act) d_out[pc] = better ? depth_tag(inb_c ? nb_d : a.depth_min, buf_c ^ 1u) : oldd_tagged;
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                if (better) nq[(q_tail + rank) & (NQ - 1)] = propagate_entry(pc, buf_c, pn, inb_c, depth_buffer(nb_tagged));
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    propagate_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1);
                    q_head += AMVS_WAVE;
                }
            }
        } else {
            float delta = (rng_uniform(h0c) * 2.0f - 1.0f) * a.depth_range;
            float d = oldd + delta;
            d = d < a.depth_min ? a.depth_min : d;
            d = d > a.depth_max ? a.depth_max : d;
            if (act) d_out[pc] = better ? depth_tag(d, buf_c) : oldd_tagged;
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                if (better) nq[(q_tail + rank) & (NQ - 1)] = make_uint2((unsigned)pc | (buf_c << 31), h0c);
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    refine_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, a.normal_range);
                    q_head += AMVS_WAVE;
                }
            }
        }
    }
    if (mode == MODE_REFINE || mode == MODE_PROP) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            if (mode == MODE_REFINE) refine_normals(nq, q_head, n, lane, nbuf0, nbuf1, a.normal_range);
            else propagate_normals(nq, q_head, n, lane, nbuf0, nbuf1);
            q_head += n;
        }
    }
}

// ------------------------------------------------------------------ plane sweep --
// _plane_sweep_torch (dense_stereo.py:262-310) in the fast arithmetic; structure (strips, plane
// chunks, 16-bit running-best keys in LDS, atomicMax merge) as plane_sweep_kernel.
template <int K, int S>
__global__ __launch_bounds__(AMVS_WAVE) void plane_sweep_fast_kernel(const SweepArgs a)
{
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    constexpr int NL = FRing<S>::NL;
    __shared__ uint16_t best[AMVS_SWEEP_MAX_TH][AMVS_WAVE];
    __shared__ float lring[(NL > 0 ? NL : 1) * K * AMVS_WAVE];

    const int lane = threadIdx.x;
    const int t0 = xcd_remap(blockIdx.x, gridDim.x);
    const int cid = t0 % a.n_chunks;
    const int t = t0 / a.n_chunks;
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    const int job_id = t / tiles_per_job;
    const int rem = t - job_id * tiles_per_job;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int d_begin = cid * a.chunk, d_end = min(a.D, d_begin + a.chunk);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    constexpr int PADW = 2 * AMVS_PAIR_BORDER;
    // (global address space: a generic pointer would make these FLAT loads, which force vmcnt(0) and
    // lgkmcnt(0) waits)
    const GlobalU16 ref_pairs = (GlobalU16)job->ref_pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    const FastConsts fc = make_fast_consts(H, W, 0);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const float fx = (float)xr;
    FastCol cols[S];
    fast_columns<S>(job, fx, cols);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int trows = min(a.TH, H - y0);
    const int rows = trows + 2 * HALF;

    for (int i = 0; i < trows; ++i) best[i][lane] = (uint16_t)0;

    // The lean reciprocal (v_rcp_f32 + one FMA) equals 1.0f / z wherever 2^-95 <= |z| < 2^96
    // (amvs_device.h).  For a plane, z = fma(depth, fma(M7, y, t2), b2) + 1e-8 is monotonic along a
    // lane's column, so the test is made ONCE per strip and plane at the strip's first and last row
    // (same sign at both ends: no zero crossing inside) instead of in every row; a strip that fails runs
    // its rows with the IEEE quotient -- the same values either way (measured +2 %: 61.7 against 60.5
    // G px-hyp/s in one run).
    const float fy_first = (float)(y0 - HALF), fy_last = (float)(y0 - HALF + rows - 1);

    for (int d = d_begin; d < d_end; ++d) {
        const float depth = a.depths[d];
        bool lean_ok = true;
        {
            JobCP jr = reload(job);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float m7 = jr->fsrc[s].M[7], b2 = jr->fsrc[s].b[2];
                const float z0 = __builtin_fmaf(depth, __builtin_fmaf(m7, fy_first, cols[s].t2), b2) + 1e-8f;
                const float z1 = __builtin_fmaf(depth, __builtin_fmaf(m7, fy_last, cols[s].t2), b2) + 1e-8f;
                const float a0 = __builtin_fabsf(z0), a1 = __builtin_fabsf(z1);
                lean_ok &= (a0 >= 0x1p-95f) & (a0 < 0x1p96f) & (a1 >= 0x1p-95f) & (a1 < 0x1p96f) & ((z0 > 0.0f) == (z1 > 0.0f));
            }
        }
        const bool lean_strip = __all(lean_ok);
        uint32_t rb[RefBytes<K>::NB];
        float ring_v[FRing<S>::NR][K];
        typename Hist<K, S>::T hist_ok = 0;
#pragma unroll
        for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int s = 0; s < FRing<S>::NR; ++s) ring_v[s][i] = 0.0f;
        int wslot = 0;

        for (int r = 0; r < rows; ++r) {
            const int yr = y0 - HALF + r;
            const bool live = col_in & ((unsigned)yr < (unsigned)H);
            const int pix = yr * W + xr;
            const uint32_t rc_raw = ref_pairs[live ? pix + PADW * yr : 0];
            const uint32_t rcode = live ? (rc_raw & 0xFFu) : 0u;
            float v[S];
            bool unused_ok = true;
            const unsigned okbits = lean_strip
                ? fast_sample_sources<S, true, false, false>(job, fc, cols, (float)yr, depth, live, v, unused_ok)
                : fast_sample_sources<S, false, false, false>(reload(job), fc, cols, (float)yr, depth, live, v, unused_ok);
            ref_bytes_push<K>(rb, rcode);
            fring_push<K, S>(lring, lane, wslot, ring_v, v);
            wslot = wslot + 1 == K ? 0 : wslot + 1;
            hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
            if (r < 2 * HALF) continue;

            const int yc = yr - HALF;
            const int xc = xr + HALF;
            const bool outl = (lane < OUTW) & (xc < W);
            const f32x2_t mv1 = ref_stats[outl ? yc * W + xc : 0];
            const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);
            float rr[K];
#pragma unroll
            for (int i = 0; i < K; ++i) rr[i] = ref_bytes_get<K>(rb, i);
            float bvs[S], bvvs[S], brvs[S];
            window_sums_fast<K, S>(lring, wslot, rr, ring_v, lane, bvs, bvvs, brvs);
            const float m1 = mv1.x, v1 = mv1.y;
            uint32_t votes = 0u;
            if (a.thresh > 0.0f) {
                // ncc > thresh (dense_stereo.py:303) without square root and division:
                // cov / sqrt(x) > t  <=>  cov > 0, x >= 0 (a negative x is the reference's NaN) and cov^2 > t^2 x
                const float t2 = a.thresh * a.thresh;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float mean2 = bvs[s] * C1;
                    const float var2 = __builtin_fmaf(-mean2, mean2, bvvs[s] * C2);
                    const float cov = __builtin_fmaf(-m1, mean2, brvs[s] * C2);
                    const float x = v1 * var2 + 1e-8f;
                    const bool vote = (cov > 0.0f) & (x >= 0.0f) & (cov * cov > t2 * x) & (((okc >> s) & 1u) != 0u);
                    votes += vote ? 1u : 0u;
                }
            } else {
                auto vote_stage = [&](auto lean, bool &ok) {
                    constexpr bool LEAN = decltype(lean)::value;
                    votes = 0u;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        // _compute_ncc_torch: eps inside the sqrt (dense_stereo.py:344-345)
                        const float mean2 = bvs[s] * C1;
                        const float var2 = __builtin_fmaf(-mean2, mean2, bvvs[s] * C2);
                        const float cov = __builtin_fmaf(-m1, mean2, brvs[s] * C2);
                        const float den = sqrt_t<LEAN>(v1 * var2 + 1e-8f, ok);
                        const float ncc = cov * rcp_t<LEAN>(den, ok);
                        if (ncc > a.thresh && ((okc >> s) & 1u)) votes += 1u;   // :303-304
                    }
                };
                bool ok = true;
                vote_stage(std::true_type{}, ok);
                if (__builtin_expect(!__all(ok), 0)) vote_stage(std::false_type{}, ok);
            }
            if (outl) {
                const uint32_t keyv = (votes << 12) | (uint32_t)(AMVS_SWEEP_MAX_CHUNK - 1 - (d - d_begin));
                const uint32_t cur = best[yc - y0][lane];
                if (keyv > cur) best[yc - y0][lane] = (uint16_t)keyv;
            }
        }
    }

    unsigned *__restrict__ keys = a.keys + job->slot * HW;
    const int xc = xr + HALF;
    if (lane < OUTW && xc < W)
        for (int i = 0; i < trows; ++i) {
            const uint32_t b = best[i][lane];
            const uint32_t plane = (uint32_t)d_begin + (AMVS_SWEEP_MAX_CHUNK - 1 - (b & (AMVS_SWEEP_MAX_CHUNK - 1)));
            atomicMax(&keys[(y0 + i) * W + xc], ((b >> 12) << 16) | (65535u - plane));
        }
}

// ------------------------------------------------------------------ sample dump ---
// Test hook (amvs_sample_sources): the sampled value of every pixel in every source at the
// pixel's own depth, in code units, and the validity bits -- the stage before the box filter.
template <int S>
__global__ __launch_bounds__(AMVS_WAVE) void sample_dump_fast_kernel(const StepArgs a, float *__restrict__ out,
                                                                     unsigned char *__restrict__ valid_out)
{
    const JobCP job = (JobCP)a.jobs;
    const int H = a.H, W = a.W;
    const FastConsts fc = make_fast_consts(H, W, a.mode == MODE_EVAL ? a.TH : 0);   // TH carries k/2 here
    const long long HW = (long long)H * W;
    const int x = blockIdx.x * AMVS_WAVE + threadIdx.x, y = blockIdx.y;
    const bool live = x < W;
    const float d = a.d_in[live ? y * W + x : 0];
    float v[S];
    unsigned okbits;
    FastCol cols[S];
    fast_columns<S>(job, (float)x, cols);
    if (a.mode == MODE_EVAL + 100) okbits = fast_sample_sources_checked<S, false>(job, fc, cols, (float)y, d, live, v);
    else okbits = fast_sample_sources_checked<S, true>(job, fc, cols, (float)y, d, live, v);
    if (live) {
#pragma unroll
        for (int s = 0; s < S; ++s) out[s * HW + y * W + x] = v[s];
        valid_out[y * W + x] = (unsigned char)okbits;
    }
}

template <int S>
static hipError_t launch_sample_dump_fast_s(const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st)
{
    hipLaunchKernelGGL((sample_dump_fast_kernel<S>), dim3((a.W + AMVS_WAVE - 1) / AMVS_WAVE, a.H), dim3(AMVS_WAVE), 0, st,
                       a, out, valid_out);
    return hipGetLastError();
}

hipError_t launch_sample_dump_fast(int S, const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st)
{
    switch (S) {
    case 2: return launch_sample_dump_fast_s<2>(a, out, valid_out, st);
    case 3: return launch_sample_dump_fast_s<3>(a, out, valid_out, st);
    case 4: return launch_sample_dump_fast_s<4>(a, out, valid_out, st);
    case 5: return launch_sample_dump_fast_s<5>(a, out, valid_out, st);
    case 6: return launch_sample_dump_fast_s<6>(a, out, valid_out, st);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------ ref stats ----
// (mean1, var1) of mvs_patchmatch.py:403,406 from the 8-bit codes: exact integer window sums
// (<= k^2 * 255^2 < 2^24), then mean1 = sum * C1, var1 = fma(-mean1, mean1, sumsq * C2).
template <int K>
__global__ __launch_bounds__(256) void fast_stats_kernel(const uint16_t *__restrict__ pairs, int H, int W,
                                                         float2 *__restrict__ out)
{
    constexpr int HALF = K / 2, B = AMVS_PAIR_BORDER;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    const int PW = W + 2 * B;
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        int sr = 0, srr = 0;
        for (int dy = -HALF; dy <= HALF; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = -HALF; dx <= HALF; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const int c = pairs[(long long)(yy + B) * PW + xx + B] & 0xFF;
                sr += c; srr += c * c;
            }
        }
        const float m1 = (float)sr * C1;
        out[i] = make_float2(m1, __builtin_fmaf(-m1, m1, (float)srr * C2));
    }
}

hipError_t launch_fast_stats(int K, const uint16_t *pairs_view, int H, int W, float2 *out, hipStream_t st)
{
    const long long n = (long long)H * W;
    const dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), blk(256);
    switch (K) {
    case 3: hipLaunchKernelGGL((fast_stats_kernel<3>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 5: hipLaunchKernelGGL((fast_stats_kernel<5>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 7: hipLaunchKernelGGL((fast_stats_kernel<7>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 9: hipLaunchKernelGGL((fast_stats_kernel<9>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 11: hipLaunchKernelGGL((fast_stats_kernel<11>), grid, blk, 0, st, pairs_view, H, W, out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ host side ----
// M = K R_s R_ref^T K^-1, b = K (t_s - R_s R_ref^T t_ref): double arithmetic on the float32
// operands, sums left to right, K^-1 by cofactors, one rounding to float32 at the end.
void fast_compose(const float K[9], const float Rr[9], const float tr[3], const float Rs[9], const float ts[3],
                  float M[9], float b[3])
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

// ------------------------------------------------------------------ dispatch -----
template <int K, int S>
static hipError_t launch_step_fast_ks(const StepArgs &a, int nblk, hipStream_t st)
{
    const int nwg = (nblk + AMVS_WG_WAVES - 1) / AMVS_WG_WAVES;
    const dim3 grid(nwg), block(AMVS_WAVE * AMVS_WG_WAVES);
    if (a.presampled) {
        if (!a.samples) return hipErrorInvalidValue;
        if (a.mode == MODE_REFINE) hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_REFINE, true>), grid, block, 0, st, a);
        else if (a.mode == MODE_PROP) hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_PROP, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pm_step_fast_kernel<K, S, -1, true>), grid, block, 0, st, a);
        return hipGetLastError();
    }
    if constexpr (fast_pair_supported(K, S)) {
        if (a.paired && (a.mode == MODE_REFINE || a.mode == MODE_PROP)) {
            const int pwg = a.n_jobs * ((a.tiles_x + AMVS_PAIR_COLS - 1) / AMVS_PAIR_COLS) * ((a.tiles_y + 1) / 2);
            const unsigned PXL = StepLds<K, S>::extra(a.wg_cap, true);
            const dim3 pblock(AMVS_WAVE * PAIR_WAVES);
            if (a.mode == MODE_REFINE)
                hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_REFINE, false, true>), dim3(pwg), pblock, PXL, st, a);
            else
                hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_PROP, false, true>), dim3(pwg), pblock, PXL, st, a);
            return hipGetLastError();
        }
    }
    const unsigned XL = StepLds<K, S>::extra(a.wg_cap);
    if (a.mode == MODE_REFINE) hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_REFINE>), grid, block, XL, st, a);
    else if (a.mode == MODE_PROP) hipLaunchKernelGGL((pm_step_fast_kernel<K, S, MODE_PROP>), grid, block, XL, st, a);
    else hipLaunchKernelGGL((pm_step_fast_kernel<K, S, -1>), grid, block, XL, st, a);
    return hipGetLastError();
}

template <int K, int S>
static hipError_t launch_sweep_fast_ks(const SweepArgs &a, int nblk, hipStream_t st)
{
    hipLaunchKernelGGL((plane_sweep_fast_kernel<K, S>), dim3(nblk), dim3(AMVS_WAVE), 0, st, a);
    return hipGetLastError();
}

template <int K, int S>
static int step_fast_occupancy_ks(int wg_cap)
{
    int n = 0;
    constexpr int TPB = AMVS_WAVE * AMVS_WG_WAVES;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pm_step_fast_kernel<K, S, MODE_REFINE>, TPB,
                                                                StepLds<K, S>::extra(wg_cap));
    return e == hipSuccess && n > 0 ? n * AMVS_WG_WAVES : 8;
}

#define AMVS_FOR_S(K, FN, ...)                                      \
    switch (S) {                                                    \
    case 2: return FN<K, 2>(__VA_ARGS__);                           \
    case 3: return FN<K, 3>(__VA_ARGS__);                           \
    case 4: return FN<K, 4>(__VA_ARGS__);                           \
    case 5: return FN<K, 5>(__VA_ARGS__);                           \
    case 6: return FN<K, 6>(__VA_ARGS__);                           \
    default: return decltype(FN<K, 2>(__VA_ARGS__))(1);             \
    }

int step_fast_waves_per_cu(int K, int S, int wg_cap)
{
    switch (K) {
    case 3: AMVS_FOR_S(3, step_fast_occupancy_ks, wg_cap)
    case 5: AMVS_FOR_S(5, step_fast_occupancy_ks, wg_cap)
    case 7: AMVS_FOR_S(7, step_fast_occupancy_ks, wg_cap)
    case 9: AMVS_FOR_S(9, step_fast_occupancy_ks, wg_cap)
    case 11: AMVS_FOR_S(11, step_fast_occupancy_ks, wg_cap)
    default: return 8;
    }
}

bool step_fast_pair_supported(int K, int S) { return patch_supported(K) && S >= 2 && S <= AMVS_KMAX_SRC && fast_pair_supported(K, S); }

hipError_t launch_step_fast(int K, int S, const StepArgs &a, hipStream_t st)
{
    if (!a.pairs) return hipErrorInvalidValue;        // fast mode samples the packed 8-bit maps only
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y;
    switch (K) {
    case 3: AMVS_FOR_S(3, launch_step_fast_ks, a, nblk, st)
    case 5: AMVS_FOR_S(5, launch_step_fast_ks, a, nblk, st)
    case 7: AMVS_FOR_S(7, launch_step_fast_ks, a, nblk, st)
    case 9: AMVS_FOR_S(9, launch_step_fast_ks, a, nblk, st)
    case 11: AMVS_FOR_S(11, launch_step_fast_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

// Resident workgroups of the sampling kernel per CU when it runs alone, through unused dynamic LDS;
// the remainder of the 160 KiB is what the window kernel of another view group can take beside it.
#ifndef AMVS_SAMPLE_LDS_BYTES
#define AMVS_SAMPLE_LDS_BYTES (27 * 1024)
#endif

template <int S>
static hipError_t launch_sample_fast_s(const StepArgs &a, hipStream_t st)
{
    const int nblk = a.n_jobs * a.s_tiles_x * a.s_tiles_y;
    const dim3 grid((nblk + AMVS_WG_WAVES - 1) / AMVS_WG_WAVES), block(AMVS_WAVE * AMVS_WG_WAVES);
    const unsigned XL = a.s_lds > 0 ? (unsigned)a.s_lds : (unsigned)AMVS_SAMPLE_LDS_BYTES;
    if (a.mode == MODE_REFINE) hipLaunchKernelGGL((pm_sample_fast_kernel<S, MODE_REFINE>), grid, block, XL, st, a);
    else if (a.mode == MODE_PROP) hipLaunchKernelGGL((pm_sample_fast_kernel<S, MODE_PROP>), grid, block, XL, st, a);
    else hipLaunchKernelGGL((pm_sample_fast_kernel<S, -1>), grid, block, XL, st, a);
    return hipGetLastError();
}

hipError_t launch_sample_fast(int S, const StepArgs &a, hipStream_t st)
{
    if (!a.pairs || !a.samples || a.s_TH < 1) return hipErrorInvalidValue;
    switch (S) {
    case 2: return launch_sample_fast_s<2>(a, st);
    case 3: return launch_sample_fast_s<3>(a, st);
    case 4: return launch_sample_fast_s<4>(a, st);
    case 5: return launch_sample_fast_s<5>(a, st);
    case 6: return launch_sample_fast_s<6>(a, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sweep_fast(int K, int S, const SweepArgs &a, hipStream_t st)
{
    if (!a.pairs) return hipErrorInvalidValue;
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y * a.n_chunks;
    switch (K) {
    case 3: AMVS_FOR_S(3, launch_sweep_fast_ks, a, nblk, st)
    case 5: AMVS_FOR_S(5, launch_sweep_fast_ks, a, nblk, st)
    case 7: AMVS_FOR_S(7, launch_sweep_fast_ks, a, nblk, st)
    case 9: AMVS_FOR_S(9, launch_sweep_fast_ks, a, nblk, st)
    case 11: AMVS_FOR_S(11, launch_sweep_fast_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

}  // namespace amvs
