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
#define AMVS_TU_ID 2
#include "amvs_fast_common.h"

namespace amvs {

// the reference view's packed map is addressed from its pixel (0,0): valid texel indices [-origin, elems - origin)
#define AMVS_REF_PAIR_IDX(i) AMVS_IDX_LOHI((i), -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))

template <int K, int S> struct StepLds {
    static constexpr unsigned PER_WAVE = (FRing<K, S>::NL > 0 ? FRing<K, S>::NL : 1) * K * AMVS_WAVE * 4u + 2u * AMVS_WAVE * 4u;
    static constexpr unsigned STATIC = AMVS_WG_WAVES * PER_WAVE;
    // paired bands: the exchange rows of a wave -- only the sources whose ring lives in registers; the others are
    // read straight from the partner's LDS ring
    static constexpr unsigned XBUF = (K / 2) * (S - FRing<K, S>::NL > 0 ? S - FRing<K, S>::NL : 0) * AMVS_WAVE * 4u;
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
// (round 4: every compiled patch size -- a band's rows of the LDS-resident sources are read from the partner's
// ring itself, only the register-resident source goes through exchange rows: 11 x 11, S = 4: 8 448 B of rings +
// 512 B of queue + 1 280 B of exchange rows = exactly the 10 240 B a wave may take at four workgroups per CU)
constexpr bool fast_pair_supported(int K, int S)
{
    return K <= 11 && S <= 4;
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
        const float d_raw = d_in[AMVS_IDX(inb ? pix + noff : 0, HW)];
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        const float dc = candidate_depth(a, mode, inb, d_raw, h0);
        float v[S];
        const unsigned okbits = fast_sample_sources_checked<S, true>(job, fc, cols, (float)yr, dc, live, v);
        if (live) {
#pragma unroll
            for (int s = 0; s < S; ++s) out[AMVS_IDX(s * HW + pix, S * HW)] = sample_encode(v[s], (okbits >> s) & 1u);
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
    constexpr int NL = FRing<K, S>::NL;
    __shared__ float lring_all[WGW * (NL > 0 ? NL : 1) * K * AMVS_WAVE];
    constexpr int NQ = 2 * AMVS_WAVE;
    __shared__ uint32_t nq_all[WGW * NQ];
    // Paired bands: a wave finishes its last K/2 output rows from the samples of the partner band's K/2 rows
    // next to the boundary.  For the NL sources whose ring lives in LDS those rows ARE in the partner's ring when
    // the partners meet (its K newest rows; the slots it overwrites afterwards hold its K/2 + 1 OLDEST rows), so
    // they are read from there; only the XS sources with register rings go through exchange rows.
    constexpr int XS = S - NL > 0 ? S - NL : 0;
    __shared__ float xbuf_all[PAIR && XS > 0 ? WGW * HALF * XS * AMVS_WAVE : 1];  // [wave][row][register source][lane]

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    float *lring = lring_all + wv * ((NL > 0 ? NL : 1) * K * AMVS_WAVE);
    uint32_t *nq = nq_all + wv * NQ;
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
#ifdef AMVS_CHECK_SELFTEST
    // (proof that the index-checked build reports: one deliberately out-of-range index per launch, never dereferenced)
    if (blockIdx.x == 0 && threadIdx.x == 0) (void)AMVS_IDX(HW + 7, HW);
#endif
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
    float *xmine = PAIR ? xbuf_all + wv * (HALF * XS * AMVS_WAVE) : nullptr;
    const float *xpartner = PAIR ? xbuf_all + (wv ^ AMVS_PAIR_COLS) * (HALF * XS * AMVS_WAVE) : nullptr;
    const float *lring_p = PAIR ? lring_all + (wv ^ AMVS_PAIR_COLS) * ((NL > 0 ? NL : 1) * K * AMVS_WAVE) : nullptr;
    // ring slot of the partner's own row next to the boundary (its step n_own_p - 1; a step's slot is step mod K)
    int pslot = 0;
    if (PAIR && paired) pslot = (min(a.TH, H - (ty ^ 1) * a.TH) + HALF - 1) % K;

    uint32_t rb[RefBytes<K>::NB];
    float ring_v[FRing<K, S>::NR][K];
    typename Hist<K, S>::T hist_ok = 0;
    uint32_t hist_h0[HALF + 1];
#pragma unroll
    for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int s = 0; s < FRing<K, S>::NR; ++s) ring_v[s][i] = 0.0f;
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
        const uint32_t rc_raw = ref_pairs[AMVS_REF_PAIR_IDX(live ? pix + PADW * yr : 0)];
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        float v[S];
        unsigned okbits = 0u;
        if (PAIR && !own) {
            // a row of the partner band: its samples, taken at its own candidates, from LDS (the partner
            // wrote them walking towards the boundary: the row next to it last)
            const float *xp = xpartner + (HALF - 1 - (loc - n_own)) * (XS * AMVS_WAVE);
#pragma unroll
            for (int s = 0; s < S; ++s)
                v[s] = s < NL ? lring_p[(s * K + pslot) * AMVS_WAVE + lane] : xp[(s < NL ? 0 : s - NL) * AMVS_WAVE + lane];
            pslot = pslot == 0 ? K - 1 : pslot - 1;
        } else if constexpr (PRE) {
            const uint32_t *__restrict__ sp = smp + AMVS_IDX(live ? pix : 0, HW);
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
            const float d_raw = d_in[AMVS_IDX(inb ? pix + noff : 0, HW)];
            const float dc = candidate_depth(a, mode, inb, d_raw, h0);
            okbits = fast_sample_sources_checked<S, true, true>(job, fc, cols, (float)yr, dc, live, v);
            if constexpr (PAIR) {
                if (XS > 0 && paired && loc >= n_own - HALF) {  // the last K/2 own rows: for the partner
                    float *xm = xmine + (loc - (n_own - HALF)) * (XS * AMVS_WAVE);
#pragma unroll
                    for (int s = NL; s < S; ++s) xm[(s - NL) * AMVS_WAVE + lane] = v[s];
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
        const int pc = AMVS_IDX(outl ? yc * W + xc : 0, HW);
        const float oldd_tagged = d_in[pc], oldc = cost_io[pc];
        const float oldd = depth_untag(oldd_tagged, a.depth_mask);
        const unsigned buf_c = depth_buffer(oldd_tagged);     // where this pixel's current normal lives
        // propagation: the neighbour the candidate was pulled from (requested with the other state
        // loads: behind the selection it would expose a memory round trip in every row).  Lanes
        // without an output pixel have pc = 0: they must not form pc + noff, which lies BEFORE the map
        // for the negative offsets of odd iterations.
        const bool inb_c = ((unsigned)(yc + oy) < (unsigned)H) & ((unsigned)(xc + ox) < (unsigned)W);
        const int pn = AMVS_IDX((outl & inb_c) ? pc + noff : 0, HW);
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
                if (better) nq[(q_tail + rank) & (NQ - 1)] = propagate_entry(pc, buf_c, inb_c, depth_buffer(nb_tagged));
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    propagate_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, noff, (int)HW);
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
                if (better) nq[(q_tail + rank) & (NQ - 1)] = refine_entry(pc, buf_c);
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    refine_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, a.normal_range, key, (int)HW);
                    q_head += AMVS_WAVE;
                }
            }
        }
    }
    if (mode == MODE_REFINE || mode == MODE_PROP) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            if (mode == MODE_REFINE) refine_normals(nq, q_head, n, lane, nbuf0, nbuf1, a.normal_range, key, (int)HW);
            else propagate_normals(nq, q_head, n, lane, nbuf0, nbuf1, noff, (int)HW);
            q_head += n;
        }
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
    const float d = a.d_in[AMVS_IDX(live ? y * W + x : 0, HW)];
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
                const int c = pairs[AMVS_IDX((long long)(yy + B) * PW + xx + B, (long long)(H + 2 * B) * PW)] & 0xFF;
                sr += c; srr += c * c;
            }
        }
        const float m1 = (float)sr * C1;
        out[i] = make_float2(m1, __builtin_fmaf(-m1, m1, (float)srr * C2));
    }
}

hipError_t launch_fast_stats(int K, const uint16_t *pairs_view, int H, int W, float2 *out, hipStream_t st)
{
    if (!patch_compiled(K)) return launch_fast_stats_generic(K, pairs_view, H, W, out, st);
    const long long n = (long long)H * W;
    const dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), blk(256);
    switch (K) {
    case 3: hipLaunchKernelGGL((fast_stats_kernel<3>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 5: hipLaunchKernelGGL((fast_stats_kernel<5>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 7: hipLaunchKernelGGL((fast_stats_kernel<7>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 9: hipLaunchKernelGGL((fast_stats_kernel<9>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 11: hipLaunchKernelGGL((fast_stats_kernel<11>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 13: hipLaunchKernelGGL((fast_stats_kernel<13>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 15: hipLaunchKernelGGL((fast_stats_kernel<15>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 17: hipLaunchKernelGGL((fast_stats_kernel<17>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 19: hipLaunchKernelGGL((fast_stats_kernel<19>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 21: hipLaunchKernelGGL((fast_stats_kernel<21>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 23: hipLaunchKernelGGL((fast_stats_kernel<23>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 25: hipLaunchKernelGGL((fast_stats_kernel<25>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 27: hipLaunchKernelGGL((fast_stats_kernel<27>), grid, blk, 0, st, pairs_view, H, W, out); break;
    case 29: hipLaunchKernelGGL((fast_stats_kernel<29>), grid, blk, 0, st, pairs_view, H, W, out); break;
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
    case 13: AMVS_FOR_S(13, step_fast_occupancy_ks, wg_cap)
    case 15: AMVS_FOR_S(15, step_fast_occupancy_ks, wg_cap)
    case 17: AMVS_FOR_S(17, step_fast_occupancy_ks, wg_cap)
    case 19: AMVS_FOR_S(19, step_fast_occupancy_ks, wg_cap)
    case 21: AMVS_FOR_S(21, step_fast_occupancy_ks, wg_cap)
    case 23: AMVS_FOR_S(23, step_fast_occupancy_ks, wg_cap)
    case 25: AMVS_FOR_S(25, step_fast_occupancy_ks, wg_cap)
    case 27: AMVS_FOR_S(27, step_fast_occupancy_ks, wg_cap)
    case 29: AMVS_FOR_S(29, step_fast_occupancy_ks, wg_cap)
    default: return step_generic_waves_per_cu(K, S);
    }
}

bool step_fast_pair_supported(int K, int S) { return patch_compiled(K) && S >= 2 && S <= AMVS_KMAX_SRC && fast_pair_supported(K, S); }

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
    case 13: AMVS_FOR_S(13, launch_step_fast_ks, a, nblk, st)
    case 15: AMVS_FOR_S(15, launch_step_fast_ks, a, nblk, st)
    case 17: AMVS_FOR_S(17, launch_step_fast_ks, a, nblk, st)
    case 19: AMVS_FOR_S(19, launch_step_fast_ks, a, nblk, st)
    case 21: AMVS_FOR_S(21, launch_step_fast_ks, a, nblk, st)
    case 23: AMVS_FOR_S(23, launch_step_fast_ks, a, nblk, st)
    case 25: AMVS_FOR_S(25, launch_step_fast_ks, a, nblk, st)
    case 27: AMVS_FOR_S(27, launch_step_fast_ks, a, nblk, st)
    case 29: AMVS_FOR_S(29, launch_step_fast_ks, a, nblk, st)
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

}  // namespace amvs

AMVS_CHECK_TU(kernels_fast)
